// pr.hxx -- PageRank (power iteration with uniform redistribution of dangling mass).
// API parity: include/gunrock/algorithms/pr.hxx:19-265 (reference): pr::param_t
// {alpha, tol, options}, result_t{p}, problem_t{plast, iweights}, enactor_t with
// is_converged() = iteration > 0 && max|p - plast| < tol, run(...) + legacy overload.
// Paths: the engine's pull iteration (gunrock_amd/csrc/grx_pr.hip, no atomics) by
// default; the generic path below is the reference's push formulation on the
// public operators (advance over the whole graph with output `none`, so the
// source vertex comes from the load-balanced expansion instead of a binary
// search per edge as upstream's parallel_for<edge> + get_source_vertex does).
#pragma once

#include <gunrock/algorithms/algorithms.hxx>

namespace gunrock {
namespace pr {

template <typename weight_t>
struct param_t {
  weight_t alpha;
  weight_t tol;
  options_t options;
  param_t(weight_t _alpha, weight_t _tol, options_t _options = options_t())
      : alpha(_alpha), tol(_tol), options(_options) {}
};

template <typename weight_t>
struct result_t {
  weight_t* p;
  int iterations = 0;  ///< filled by run(): number of loop() executions
  result_t(weight_t* _p) : p(_p) {}
};

namespace detail {
template <typename graph_t, typename weight_t>
struct inverse_out_weight_t {
  graph_t G;
  weight_t alpha;
  weight_t* iweights;
  __host__ __device__ void operator()(typename graph_t::vertex_type const& v) const {
    weight_t sum = 0;
    const auto first = G.get_starting_edge(v);
    const auto last = first + G.get_number_of_neighbors(v);
    for (auto e = first; e < last; ++e) sum += G.get_edge_weight(e);
    iweights[v] = sum != 0 ? alpha / sum : 0;
  }
};
template <typename vertex_t, typename weight_t>
struct save_and_rebase_t {  // plast = p; p = base
  weight_t* p;
  weight_t* plast;
  const weight_t* base;
  __host__ __device__ void operator()(vertex_t const& v) const {
    plast[v] = p[v];
    p[v] = *base;
  }
};
template <typename vertex_t, typename edge_t, typename weight_t>
struct spread_t {
  weight_t* p;
  const weight_t* plast;
  const weight_t* iweights;
  __host__ __device__ bool operator()(vertex_t const& src, vertex_t const& dst, edge_t const&,
                                      weight_t const& w) const {
    math::atomic::add(p + dst, plast[src] * iweights[src] * w);
    return false;
  }
};
// block-free reductions through a single atomic per element are fine off the
// fast path; the generic path favours clarity.
template <typename vertex_t, typename weight_t>
struct dangling_t {
  const weight_t* p;
  const weight_t* iweights;
  weight_t alpha;
  weight_t* sum;
  __host__ __device__ void operator()(vertex_t const& v) const {
    if (iweights[v] == 0) math::atomic::add(sum, alpha * p[v]);
  }
};
template <typename vertex_t, typename weight_t>
struct max_change_t {
  const weight_t* p;
  const weight_t* plast;
  weight_t* err;
  __host__ __device__ void operator()(vertex_t const& v) const {
    const weight_t d = p[v] > plast[v] ? p[v] - plast[v] : plast[v] - p[v];
    if (d > 0) math::atomic::max(err, d);
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

  vector_t<weight_t, memory_space_t::device> plast;
  vector_t<weight_t, memory_space_t::device> iweights;
  vector_t<weight_t, memory_space_t::device> scalars;  // [0] dangling sum, [1] base, [2] max change

  problem_t(graph_t& G, param_type& _param, result_type& _result,
            std::shared_ptr<gcuda::multi_context_t> _context)
      : gunrock::problem_t<graph_t>(G, _context), param(_param), result(_result) {}

  void init() override {
    const std::size_t n = (std::size_t)this->get_graph().get_number_of_vertices();
    plast.resize(n);
    iweights.resize(n);
    scalars.resize(4);
  }

  void reset() override {
    auto g = this->get_graph();
    const std::size_t n = (std::size_t)g.get_number_of_vertices();
    auto stream = this->get_single_context()->stream();
    const unsigned grid = frontier::detail::grid_for(n);
    hipLaunchKernelGGL((frontier::detail::fill_kernel<weight_t>), dim3(grid), dim3(256), 0, stream, result.p,
                       (weight_t)(1.0 / (double)n), n);
    hipLaunchKernelGGL((frontier::detail::fill_kernel<weight_t>), dim3(grid), dim3(256), 0, stream,
                       memory::raw_pointer_cast(plast.data()), (weight_t)0, n);
    detail::inverse_out_weight_t<graph_t, weight_t> op{g, param.alpha, memory::raw_pointer_cast(iweights.data())};
    operators::parallel_for::execute<operators::parallel_for_each_t::vertex>(g, op, *this->context);
  }
};

template <typename problem_t>
struct enactor_t : gunrock::enactor_t<problem_t> {
  using vertex_t = typename problem_t::vertex_t;
  using edge_t = typename problem_t::edge_t;
  using weight_t = typename problem_t::weight_t;

  enactor_t(problem_t* _problem, std::shared_ptr<gcuda::multi_context_t> _context,
            enactor_properties_t _properties)
      : gunrock::enactor_t<problem_t>(_problem, _context, _properties) {}

  void loop(gcuda::multi_context_t& context) override {
    auto E = this->get_enactor();
    auto P = this->get_problem();
    auto G = P->get_graph();
    auto* sc = context.get_context(0);
    const vertex_t n = G.get_number_of_vertices();
    weight_t* p = P->result.p;
    weight_t* plast = memory::raw_pointer_cast(P->plast.data());
    weight_t* iw = memory::raw_pointer_cast(P->iweights.data());
    weight_t* scal = memory::raw_pointer_cast(P->scalars.data());
    const weight_t alpha = P->param.alpha;
    using namespace operators;

    error::throw_if_exception(hipMemsetAsync(scal, 0, 4 * sizeof(weight_t), sc->stream()), "pr scalars");
    parallel_for::execute<parallel_for_each_t::vertex>(G, detail::dangling_t<vertex_t, weight_t>{p, iw, alpha, scal},
                                                       context);
    weight_t dsum = 0;
    error::throw_if_exception(hipMemcpyAsync(&dsum, scal, sizeof(weight_t), hipMemcpyDeviceToHost, sc->stream()),
                              "pr dsum");
    sc->synchronize();
    const weight_t base = (1 - alpha + dsum) / n;
    error::throw_if_exception(hipMemcpyAsync(scal + 1, &base, sizeof(weight_t), hipMemcpyHostToDevice, sc->stream()),
                              "pr base");
    parallel_for::execute<parallel_for_each_t::vertex>(
        G, detail::save_and_rebase_t<vertex_t, weight_t>{p, plast, scal + 1}, context);
    advance::execute<load_balance_t::merge_path, advance_direction_t::forward, advance_io_type_t::graph,
                     advance_io_type_t::none>(G, E, detail::spread_t<vertex_t, edge_t, weight_t>{p, plast, iw},
                                              context);
  }

  bool is_converged(gcuda::multi_context_t& context) override {
    if (this->iteration == 0) return false;
    auto P = this->get_problem();
    auto G = P->get_graph();
    auto* sc = context.get_context(0);
    weight_t* scal = memory::raw_pointer_cast(P->scalars.data());
    error::throw_if_exception(hipMemsetAsync(scal + 2, 0, sizeof(weight_t), sc->stream()), "pr err");
    operators::parallel_for::execute<operators::parallel_for_each_t::vertex>(
        G, detail::max_change_t<vertex_t, weight_t>{P->result.p, memory::raw_pointer_cast(P->plast.data()), scal + 2},
        context);
    weight_t err = 0;
    error::throw_if_exception(hipMemcpyAsync(&err, scal + 2, sizeof(weight_t), hipMemcpyDeviceToHost, sc->stream()),
                              "pr err");
    sc->synchronize();
    return err < P->param.tol;
  }
};

template <typename graph_t>
float run(graph_t& G, param_t<typename graph_t::weight_type>& param, result_t<typename graph_t::weight_type>& result,
          std::shared_ptr<gcuda::multi_context_t> context =
              std::shared_ptr<gcuda::multi_context_t>(new gcuda::multi_context_t(0))) {
  using weight_t = typename graph_t::weight_type;
#ifndef GUNROCK_HEADER_ONLY
  if constexpr (engine::supported_types<graph_t>()) {
    if (!(param.options.engine_flags & 1)) {
      grx_context_t ctx = engine::context_for(*context);
      grx_graph_t g = engine::graph_for(ctx, G);
      grx_options_t o = engine::to_c(param.options);
      float ms = 0.0f;
      int32_t iters = 0;
      engine::check(grx_pr(ctx, g, (float)param.alpha, (float)param.tol, &o, (float*)result.p, &iters, &ms));
      result.iterations = iters;
      auto& b = benchmark::current();
      b.search_depth = (std::size_t)iters;
      b.total_runtime = ms;
      return ms;
    }
  }
#endif
  using problem_type = problem_t<graph_t, param_t<weight_t>, result_t<weight_t>>;
  using enactor_type = enactor_t<problem_type>;
  problem_type problem(G, param, result, context);
  problem.init();
  problem.reset();
  enactor_properties_t props;
  props.self_manage_frontiers = true;
  enactor_type enactor(&problem, context, props);
  const float ms = enactor.enact();
  result.iterations = enactor.iteration;
  return ms;
}

template <typename graph_t>
float run(graph_t& G, typename graph_t::weight_type alpha, typename graph_t::weight_type tol,
          typename graph_t::weight_type* p,
          std::shared_ptr<gcuda::multi_context_t> context =
              std::shared_ptr<gcuda::multi_context_t>(new gcuda::multi_context_t(0))) {
  using weight_t = typename graph_t::weight_type;
  param_t<weight_t> param(alpha, tol);
  result_t<weight_t> result(p);
  return run(G, param, result, context);
}

}  // namespace pr
}  // namespace gunrock
