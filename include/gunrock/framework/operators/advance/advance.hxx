// advance.hxx -- the advance operator: visit the out-edges of every element of
// the input frontier, call `op(src, nbr, edge, weight)` and emit `nbr` (or the
// invalid sentinel -1) into the output frontier.
// API parity: include/gunrock/framework/operators/advance/advance.hxx:94-275
// (reference): execute<lb, direction, input_type, output_type>(G, op, in*, out*,
// segments, context); the enactor overload execute<...>(G, E, op, context,
// swap_buffers = true); execute_runtime(G, E, op, lb, context, swap_buffers).
// Same error behaviour: unsupported load balancing throws
// error::exception_t("Load balance type not supported."), multi-context throws
// "`context.size() != 1` not supported".
// What differs: warp_mapped is real, bucketing picks the kernel per frontier from a degree
// histogram (bucketing.hxx), merge_path_v2 maps onto merge_path, graph-as-input uses the row offsets as the scan (no scan launch),
// and no kernel is followed by a host synchronise.
#pragma once

#include <gunrock/cuda/context.hxx>
#include <gunrock/error.hxx>
#include <gunrock/framework/benchmark.hxx>
#include <gunrock/framework/operators/advance/block_mapped.hxx>
#include <gunrock/framework/operators/advance/bucketing.hxx>
#include <gunrock/framework/operators/advance/helpers.hxx>
#include <gunrock/framework/operators/advance/merge_path.hxx>
#include <gunrock/framework/operators/advance/thread_mapped.hxx>
#include <gunrock/framework/operators/advance/warp_mapped.hxx>
#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/util/trace.hxx>

namespace gunrock {
namespace operators {
namespace advance {

namespace detail {
// advance_direction_t::backward: the operator still sees an edge as (source, destination, edge, weight) although the walk
// starts from the destination.  The ends are forwarded as the MUTABLE lvalues the kernels hand in, so operators declared
// (vertex_t&, vertex_t&, edge_t const&, weight_t const&) -- the reference's hits.hxx:137 style -- compile in this direction
// too.  NOTE: `e` indexes the CSC arrays (the in-edge's position in its column), not the CSR arrays: an
// operator that looks a per-edge array up by CSR edge id must not be run backward.
template <typename operator_t>
struct swap_ends_t {
  operator_t op;
  template <typename vertex_t, typename edge_t, typename weight_t>
  __host__ __device__ bool operator()(vertex_t& dst, vertex_t& src, edge_t const& e, weight_t const& w) const {
    return op(src, dst, e, w);
  }
};
}  // namespace detail

template <load_balance_t lb = load_balance_t::merge_path,
          advance_direction_t direction = advance_direction_t::forward,
          advance_io_type_t input_type = advance_io_type_t::vertices,
          advance_io_type_t output_type = advance_io_type_t::vertices, typename graph_t, typename operator_t,
          typename frontier_t, typename work_tiles_t>
void execute(graph_t& G, operator_t op, frontier_t* input, frontier_t* output, work_tiles_t& segments,
             gcuda::multi_context_t& context) {
  GUNROCK_TRACE_RANGE("advance");
  using type_t = typename frontier_t::type_t;
  using edge_t = typename graph_t::edge_type;
  error::throw_if_exception(context.size() != 1, "`context.size() != 1` not supported");
  if constexpr (direction == advance_direction_t::backward) {
    // Pull: for every vertex v of the input, visit the IN-edges (u -> v) of v -- the graph's CSC view -- call
    // op(u, v, e, w) (e indexes the CSC arrays) and emit u, or -1, at the slot of that in-edge: the forward advance of the
    // reversed graph with the operator's ends swapped back, on the same load-balanced kernels.  The reference declares the
    // enum (operators/configs.hxx:78-82) and passes the template argument down without ever reading it.
    // advance_direction_t::optimized stays forward here: when to pull is the algorithm's decision (the BFS engine takes it
    // per level), not something an operator can know from a functor.
    static_assert(graph_t::has_csc_view,
                  "advance_direction_t::backward needs a graph with a CSC view: graph::build(properties, csr, csc)");
    auto R = G.reverse_view();
    execute<lb, advance_direction_t::forward, input_type, output_type>(R, detail::swap_ends_t<operator_t>{op}, input, output,
                                                                       segments, context);
    return;
  }
  static_assert(input_type != advance_io_type_t::edges && input_type != advance_io_type_t::none,
                "advance input must be `vertices` or `graph`");
  static_assert(lb != load_balance_t::work_stealing, "Load balance type not supported.");
  auto& ctx = *context.get_context(0);

  constexpr bool whole_graph = input_type == advance_io_type_t::graph;
  const type_t* in = whole_graph ? nullptr : input->data();
  const std::size_t n = whole_graph ? (std::size_t)G.get_number_of_vertices() : input->get_number_of_elements();

  std::size_t total;
  const edge_t* seg;
  int max_degree = 0;
  if constexpr (whole_graph) {
    total = (std::size_t)G.get_number_of_edges();  // the CSR offsets ARE the scan
    seg = G.get_row_offsets();
  } else {
    constexpr bool own_kernel = lb == load_balance_t::thread_mapped || lb == load_balance_t::warp_mapped || lb == load_balance_t::block_mapped;
    total = compute_output_offsets(G, input, segments, ctx, false, own_kernel ? &max_degree : nullptr);
    seg = memory::raw_pointer_cast(segments.data());
  }
  // A frontier of HUBS on a mapping that gives a row to one thread / wave / workgroup (round 6): the level behind the 125 k-edge
  // source of the LJ stand-in was ONE wave's 1.8 ms (warp_mapped; thread_mapped 0.58, block_mapped 0.51) against 16 us on the
  // merge-path kernel, the level of its neighbours 2.7 / 2.2 / 1.9 ms against 0.46 (profiles/r6_c34_*).  The output is the
  // same for every load balance (neighbour k of slot i at segments[i] + k), so such a frontier -- its longest row holds
  // >= 2048 neighbours and is either 16 x the mean row or more than a 1024th of the level -- runs on the merge-path kernel
  // whatever was asked for; the longest row comes back with the total from the scan of the degrees.
  const bool hubs = max_degree >= 2048 && ((std::size_t)max_degree * n >= 16 * total || total / (std::size_t)max_degree < 1024);
  benchmark::LOG_EDGES_HOST(total);
  benchmark::LOG_VERTICES_HOST(n);

  type_t* out = nullptr;
  if constexpr (output_type != advance_io_type_t::none) {
    if (output->get_capacity() < total) output->reserve(total);
    output->set_number_of_elements(total);
    out = output->data();
  }
  if (total == 0) return;

  if (hubs && lb != load_balance_t::merge_path && lb != load_balance_t::merge_path_v2)
    merge_path::launch<output_type>(G, op, in, n, out, seg, total, ctx);
  else if constexpr (lb == load_balance_t::thread_mapped)
    thread_mapped::launch<output_type>(G, op, in, n, out, seg, ctx);
  else if constexpr (lb == load_balance_t::warp_mapped)
    warp_mapped::launch<output_type>(G, op, in, n, out, seg, ctx);
  else if constexpr (lb == load_balance_t::block_mapped)
    block_mapped::launch<output_type>(G, op, in, n, out, seg, ctx);
  else if constexpr (lb == load_balance_t::bucketing) {
    // chosen per frontier from its degree histogram
    switch (bucketing::select(seg, n, total, ctx)) {
      case load_balance_t::thread_mapped: thread_mapped::launch<output_type>(G, op, in, n, out, seg, ctx); break;
      case load_balance_t::warp_mapped: warp_mapped::launch<output_type>(G, op, in, n, out, seg, ctx); break;
      case load_balance_t::block_mapped: block_mapped::launch<output_type>(G, op, in, n, out, seg, ctx); break;
      default: merge_path::launch<output_type>(G, op, in, n, out, seg, total, ctx);
    }
  } else
    merge_path::launch<output_type>(G, op, in, n, out, seg, total, ctx);
}

template <load_balance_t lb = load_balance_t::merge_path,
          advance_direction_t direction = advance_direction_t::forward,
          advance_io_type_t input_type = advance_io_type_t::vertices,
          advance_io_type_t output_type = advance_io_type_t::vertices, typename graph_t, typename enactor_type,
          typename operator_type>
void execute(graph_t& G, enactor_type* E, operator_type op, gcuda::multi_context_t& context,
             bool swap_buffers = true) {
  execute<lb, direction, input_type, output_type>(G, op, E->get_input_frontier(), E->get_output_frontier(),
                                                  E->scanned_work_domain, context);
  if (swap_buffers && output_type != advance_io_type_t::none) E->swap_frontier_buffers();
}

// Extension (not in the reference): advance followed by filter_algorithm_t::compact with an always-true predicate, FUSED --
// the merge-path advance compacts the neighbours `op` keeps on the fly (wave ballot, one atomic per workgroup and round), so
// the m_F-entry output with its -1 holes is never written and the filter pass never runs.  Same output SET as
// advance::execute<merge_path> + filter::execute<compact>(keep valid); order across workgroups is unspecified, as for compact.
// One size read-back instead of two.
template <advance_io_type_t output_type = advance_io_type_t::vertices, typename graph_t, typename enactor_type,
          typename operator_type>
void execute_compact(graph_t& G, enactor_type* E, operator_type op, gcuda::multi_context_t& context,
                     bool swap_buffers = true) {
  GUNROCK_TRACE_RANGE("advance+compact");
  using frontier_t = typename enactor_type::frontier_t;
  using type_t = typename frontier_t::type_t;
  using edge_t = typename graph_t::edge_type;
  error::throw_if_exception(context.size() != 1, "`context.size() != 1` not supported");
  auto& ctx = *context.get_context(0);
  frontier_t* input = E->get_input_frontier();
  frontier_t* output = E->get_output_frontier();
  const std::size_t n = input->get_number_of_elements();
  const std::size_t total = compute_output_offsets(G, input, E->scanned_work_domain, ctx, false);
  const edge_t* seg = memory::raw_pointer_cast(E->scanned_work_domain.data());
  benchmark::LOG_EDGES_HOST(total);
  benchmark::LOG_VERTICES_HOST(n);
  if (output->get_capacity() < total) output->reserve(total);
  if (total == 0) {
    output->set_number_of_elements(0);
  } else {
    int32_t* counter = ctx.template scratch<int32_t>(2, 4);
    error::throw_if_exception(hipMemsetAsync(counter, 0, sizeof(int32_t), ctx.stream()), "counter reset");
    merge_path::launch_compact<output_type>(G, op, input->data(), n, output->data(), seg, total, counter, ctx);
    output->set_number_of_elements((std::size_t)ctx.read_back(counter)[0]);
  }
  if (swap_buffers) E->swap_frontier_buffers();
}

template <typename graph_t, typename enactor_type, typename operator_type>
void execute_runtime(graph_t& G, enactor_type* E, operator_type op, load_balance_t lb,
                     gcuda::multi_context_t& context, bool swap_buffers = true) {
  constexpr auto fwd = advance_direction_t::forward;
  constexpr auto vtx = advance_io_type_t::vertices;
  switch (lb) {
    case load_balance_t::thread_mapped:
      execute<load_balance_t::thread_mapped, fwd, vtx, vtx>(G, E, op, context, swap_buffers);
      break;
    case load_balance_t::warp_mapped:
      execute<load_balance_t::warp_mapped, fwd, vtx, vtx>(G, E, op, context, swap_buffers);
      break;
    case load_balance_t::block_mapped:
      execute<load_balance_t::block_mapped, fwd, vtx, vtx>(G, E, op, context, swap_buffers);
      break;
    case load_balance_t::bucketing:
      execute<load_balance_t::bucketing, fwd, vtx, vtx>(G, E, op, context, swap_buffers);
      break;
    case load_balance_t::merge_path:
    case load_balance_t::merge_path_v2:
      execute<load_balance_t::merge_path, fwd, vtx, vtx>(G, E, op, context, swap_buffers);
      break;
    default:
      error::throw_if_exception(hipErrorUnknown, "Load balance type not supported.");
  }
}

}  // namespace advance
}  // namespace operators
}  // namespace gunrock
