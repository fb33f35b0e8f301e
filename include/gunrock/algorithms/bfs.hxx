// bfs.hxx -- breadth-first search.
// API parity: include/gunrock/algorithms/bfs.hxx:17-215 (reference): bfs::param_t,
// result_t, problem_t, enactor_t, run(G, param, result, context) and the legacy
// run(G, source, distances, predecessors, context); depth INT_MAX = unreached;
// predecessors are accepted and never written (as upstream).
//
// Two execution paths, identical depths:
//  * engine (default when libgrx is linked): the fused device-driven level loop
//    of gunrock_amd/csrc/grx_bfs.hip -- bitmap claim, compacted output, no host
//    round trip per level;
//  * generic (options.engine_flags & 1, or -DGUNROCK_HEADER_ONLY): the enactor
//    below, written against the public operators exactly like a user algorithm:
//    advance with an atomicMin relaxation, then the optional filter.
#pragma once

#include <gunrock/algorithms/algorithms.hxx>

#include <limits>

namespace gunrock {
namespace bfs {

template <typename vertex_t>
struct param_t {
  vertex_t single_source;
  options_t options;
  param_t(vertex_t _single_source, options_t _options = options_t())
      : single_source(_single_source), options(_options) {}
};

template <typename vertex_t>
struct result_t {
  vertex_t* distances;
  vertex_t* predecessors;
  result_t(vertex_t* _distances, vertex_t* _predecessors)
      : distances(_distances), predecessors(_predecessors) {}
};

namespace detail {
// Claim `nbr` for depth `next` if that improves on its label.
template <typename vertex_t, typename edge_t, typename weight_t>
struct relax_depth_t {
  vertex_t* depth;
  vertex_t next;
  __host__ __device__ bool operator()(vertex_t const&, vertex_t const& nbr, edge_t const&, weight_t const&) const {
    // A plain (possibly stale) read first: labels only decrease, so a label already at or below `next` cannot be improved
    // by this edge whatever has happened since.  The reference issues the atomic for EVERY edge (bfs.hxx:117-119); on a
    // multi-XCD part a device-scope atomic is a memory-side operation (~20 G/s for the whole chip), and > 90 % of the
    // edges of a fat level lead to labelled vertices.  Same result: the atomic still decides among the contenders.
    if (thread::load(depth + nbr) <= next) return false;
    return next < math::atomic::min(depth + nbr, next);
  }
};
template <typename vertex_t>
struct keep_valid_t {
  __host__ __device__ bool operator()(vertex_t const&) const { return true; }
};
}  // namespace detail

template <typename graph_t, typename param_type, typename result_type>
struct problem_t : gunrock::problem_t<graph_t> {
  param_type param;
  result_type result;
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;

  problem_t(graph_t& G, param_type& _param, result_type& _result,
            std::shared_ptr<gcuda::multi_context_t> _context)
      : gunrock::problem_t<graph_t>(G, _context), param(_param), result(_result) {}

  void init() override {}

  void reset() override {
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    auto stream = this->get_single_context()->stream();
    hipLaunchKernelGGL((frontier::detail::fill_kernel<vertex_t>), dim3(frontier::detail::grid_for(n)), dim3(256), 0,
                       stream, result.distances, std::numeric_limits<vertex_t>::max(), n);
    const vertex_t zero = 0;
    error::throw_if_exception(hipMemcpyAsync(result.distances + param.single_source, &zero, sizeof(vertex_t),
                                             hipMemcpyHostToDevice, stream),
                              "bfs reset");
    error::throw_if_exception(hipStreamSynchronize(stream), "bfs reset");
  }
};

template <typename problem_t>
struct enactor_t : gunrock::enactor_t<problem_t> {
  using vertex_t = typename problem_t::vertex_t;
  using edge_t = typename problem_t::edge_t;
  using weight_t = typename problem_t::weight_t;
  using frontier_t = typename gunrock::enactor_t<problem_t>::frontier_t;

  enactor_t(problem_t* _problem, std::shared_ptr<gcuda::multi_context_t> _context)
      : gunrock::enactor_t<problem_t>(_problem, _context) {}

  void prepare_frontier(frontier_t* f, gcuda::multi_context_t& context) override {
    f->push_back(this->get_problem()->param.single_source);
  }

  void loop(gcuda::multi_context_t& context) override {
    auto E = this->get_enactor();
    auto P = this->get_problem();
    auto G = P->get_graph();
    const options_t& opt = P->param.options;

    detail::relax_depth_t<vertex_t, edge_t, weight_t> relax{P->result.distances, (vertex_t)(this->iteration + 1)};
    const bool mp = opt.advance_load_balance == operators::load_balance_t::merge_path ||
                    opt.advance_load_balance == operators::load_balance_t::merge_path_v2;
    if (mp && opt.enable_filter && opt.filter_algorithm == operators::filter_algorithm_t::compact) {
      // the reference README's tuned command line (merge_path + compact filter): one fused pass, no -1 holes in HBM
      operators::advance::execute_compact(G, E, relax, context);
      return;
    }
    operators::advance::execute_runtime(G, E, relax, opt.advance_load_balance, context);
    if (opt.enable_filter)
      operators::filter::execute_runtime(G, E, detail::keep_valid_t<vertex_t>(), opt.filter_algorithm, context);
  }
};

template <typename graph_t>
float run(graph_t& G, param_t<typename graph_t::vertex_type>& param,
          result_t<typename graph_t::vertex_type>& result,
          std::shared_ptr<gcuda::multi_context_t> context =
              std::shared_ptr<gcuda::multi_context_t>(new gcuda::multi_context_t(0))) {
  using vertex_t = typename graph_t::vertex_type;
#ifndef GUNROCK_HEADER_ONLY
  if constexpr (engine::supported_types<graph_t>()) {
    if (!(param.options.engine_flags & 1)) {
      grx_context_t ctx = engine::context_for(*context);
      grx_graph_t g = engine::graph_for(ctx, G);
      grx_options_t o = engine::to_c(param.options);
      float ms = 0.0f;
      engine::check(grx_bfs(ctx, g, (int32_t)param.single_source, &o, (int32_t*)result.distances,
                            (int32_t*)result.predecessors, &ms));
      grx_run_stats_t st;
      engine::check(grx_get_run_stats(ctx, &st));
      auto& b = benchmark::current();
      b.edges_visited = (unsigned int)st.edges_visited;
      b.vertices_visited = (unsigned int)st.vertices_visited;
      b.search_depth = (std::size_t)st.search_depth;
      b.total_runtime = ms;
      return ms;
    }
  }
#endif
  using problem_type = problem_t<graph_t, param_t<vertex_t>, result_t<vertex_t>>;
  using enactor_type = enactor_t<problem_type>;
  problem_type problem(G, param, result, context);
  problem.init();
  problem.reset();
  enactor_type enactor(&problem, context);
  return enactor.enact();
}

template <typename graph_t>
float run(graph_t& G, typename graph_t::vertex_type& single_source, typename graph_t::vertex_type* distances,
          typename graph_t::vertex_type* predecessors,
          std::shared_ptr<gcuda::multi_context_t> context =
              std::shared_ptr<gcuda::multi_context_t>(new gcuda::multi_context_t(0))) {
  using vertex_t = typename graph_t::vertex_type;
  param_t<vertex_t> param(single_source);
  result_t<vertex_t> result(distances, predecessors);
  return run(G, param, result, context);
}

}  // namespace bfs
}  // namespace gunrock
