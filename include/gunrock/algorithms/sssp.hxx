// sssp.hxx -- single-source shortest paths.
// API parity: include/gunrock/algorithms/sssp.hxx:17-230 (reference): sssp::param_t,
// result_t(distances, predecessors, n_vertices), problem_t (visited stamps),
// enactor_t, run(...) + legacy overload.  Unreached = numeric_limits<weight_t>::max()
// (FLT_MAX, not inf).  Label-correcting search; for non-negative weights the
// fixed point equals Dijkstra's result bit for bit.
// Paths: fused engine (gunrock_amd/csrc/grx_sssp.hip) by default, generic
// operators below with options.engine_flags & 1 or -DGUNROCK_HEADER_ONLY.
#pragma once

#include <gunrock/algorithms/algorithms.hxx>

#include <limits>

namespace gunrock {
namespace sssp {

template <typename vertex_t>
struct param_t {
  vertex_t single_source;
  options_t options;
  param_t(vertex_t _single_source, options_t _options = options_t())
      : single_source(_single_source), options(_options) {}
};

template <typename vertex_t, typename weight_t>
struct result_t {
  weight_t* distances;
  vertex_t* predecessors;
  result_t(weight_t* _distances, vertex_t* _predecessors, vertex_t n_vertices = 0)
      : distances(_distances), predecessors(_predecessors) {}
};

namespace detail {
template <typename vertex_t, typename edge_t, typename weight_t>
struct relax_distance_t {
  weight_t* dist;
  __host__ __device__ bool operator()(vertex_t const& src, vertex_t const& nbr, edge_t const&,
                                      weight_t const& w) const {
    const weight_t through = thread::load(dist + src) + w;
    return through < math::atomic::min(dist + nbr, through);
  }
};
// keep a vertex once per iteration (stamp taken atomically, unlike upstream's
// racy load/store pair, sssp.hxx:134-137)
template <typename vertex_t>
struct once_per_iteration_t {
  vertex_t* stamp;
  vertex_t iteration;
  __host__ __device__ bool operator()(vertex_t const& v) const {
    return math::atomic::exch(stamp + v, iteration) != iteration;
  }
};
}  // namespace detail

template <typename graph_t, typename param_type, typename result_type>
struct problem_t : gunrock::problem_t<graph_t> {
  param_type param;
  result_type result;
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;

  vector_t<vertex_t, memory_space_t::device> visited;

  problem_t(graph_t& G, param_type& _param, result_type& _result,
            std::shared_ptr<gcuda::multi_context_t> _context)
      : gunrock::problem_t<graph_t>(G, _context), param(_param), result(_result) {}

  void init() override { visited.resize((std::size_t)this->get_graph().get_number_of_vertices()); }

  void reset() override {
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    auto stream = this->get_single_context()->stream();
    const unsigned g = frontier::detail::grid_for(n);
    hipLaunchKernelGGL((frontier::detail::fill_kernel<weight_t>), dim3(g), dim3(256), 0, stream, result.distances,
                       std::numeric_limits<weight_t>::max(), n);
    hipLaunchKernelGGL((frontier::detail::fill_kernel<vertex_t>), dim3(g), dim3(256), 0, stream,
                       memory::raw_pointer_cast(visited.data()), (vertex_t)-1, n);
    const weight_t zero = 0;
    error::throw_if_exception(hipMemcpyAsync(result.distances + param.single_source, &zero, sizeof(weight_t),
                                             hipMemcpyHostToDevice, stream),
                              "sssp reset");
    error::throw_if_exception(hipStreamSynchronize(stream), "sssp reset");
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

    detail::relax_distance_t<vertex_t, edge_t, weight_t> relax{P->result.distances};
    detail::once_per_iteration_t<vertex_t> once{memory::raw_pointer_cast(P->visited.data()),
                                                (vertex_t)this->iteration};
    operators::advance::execute_runtime(G, E, relax, opt.advance_load_balance, context);
    operators::filter::execute<operators::filter_algorithm_t::bypass>(G, E, once, context);
    if (opt.enable_uniquify)
      operators::uniquify::execute<operators::uniquify_algorithm_t::unique>(E, context, opt.best_effort_uniquify,
                                                                            opt.uniquify_percent);
  }
};

template <typename graph_t>
float run(graph_t& G, param_t<typename graph_t::vertex_type>& param,
          result_t<typename graph_t::vertex_type, typename graph_t::weight_type>& result,
          std::shared_ptr<gcuda::multi_context_t> context =
              std::shared_ptr<gcuda::multi_context_t>(new gcuda::multi_context_t(0))) {
  using vertex_t = typename graph_t::vertex_type;
  using weight_t = typename graph_t::weight_type;
#ifndef GUNROCK_HEADER_ONLY
  if constexpr (engine::supported_types<graph_t>()) {
    if (!(param.options.engine_flags & 1)) {
      grx_context_t ctx = engine::context_for(*context);
      grx_graph_t g = engine::graph_for(ctx, G);
      grx_options_t o = engine::to_c(param.options);
      float ms = 0.0f;
      engine::check(grx_sssp(ctx, g, (int32_t)param.single_source, &o, (float*)result.distances,
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
  using problem_type = problem_t<graph_t, param_t<vertex_t>, result_t<vertex_t, weight_t>>;
  using enactor_type = enactor_t<problem_type>;
  problem_type problem(G, param, result, context);
  problem.init();
  problem.reset();
  enactor_type enactor(&problem, context);
  return enactor.enact();
}

template <typename graph_t>
float run(graph_t& G, typename graph_t::vertex_type& single_source, typename graph_t::weight_type* distances,
          typename graph_t::vertex_type* predecessors,
          std::shared_ptr<gcuda::multi_context_t> context =
              std::shared_ptr<gcuda::multi_context_t>(new gcuda::multi_context_t(0))) {
  using vertex_t = typename graph_t::vertex_type;
  using weight_t = typename graph_t::weight_type;
  param_t<vertex_t> param(single_source);
  result_t<vertex_t, weight_t> result(distances, predecessors, G.get_number_of_vertices());
  return run(G, param, result, context);
}

}  // namespace sssp
}  // namespace gunrock
