// for.hxx -- parallel_for: apply `op` to every vertex / edge / edge weight of the
// graph, or to every valid element of a frontier.
// API parity: include/gunrock/framework/operators/for/for.hxx:25-107 (reference),
// which is thrust::for_each over a counting iterator; here a grid-stride kernel.
#pragma once

#include <gunrock/cuda/context.hxx>
#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/util/type_limits.hxx>

namespace gunrock {
namespace operators {
namespace parallel_for {
namespace detail {

template <parallel_for_each_t type, typename graph_t, typename func_t, typename index_t>
__global__ __launch_bounds__(256) void graph_kernel(graph_t G, func_t op, std::size_t n) {
  for (std::size_t i = (std::size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (std::size_t)gridDim.x * 256) {
    if constexpr (type == parallel_for_each_t::weight)
      op(G.get_edge_weight((index_t)i));
    else
      op((index_t)i);
  }
}

template <typename type_t, typename func_t>
__global__ __launch_bounds__(256) void element_kernel(const type_t* p, func_t op, std::size_t n) {
  for (std::size_t i = (std::size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (std::size_t)gridDim.x * 256) {
    const type_t x = p[i];
    if (gunrock::util::limits::is_valid(x)) op(x);
  }
}

inline unsigned grid(std::size_t n, gcuda::standard_context_t& ctx) {
  std::size_t g = (n + 255) / 256;
  const std::size_t cap = (std::size_t)ctx.props().multiProcessorCount * 16;
  return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace detail

template <parallel_for_each_t type, typename func_t, typename frontier_t>
std::enable_if_t<type == parallel_for_each_t::element> execute(frontier_t& f, func_t op,
                                                               gcuda::multi_context_t& context) {
  using type_t = typename frontier_t::type_t;
  auto& ctx = *context.get_context(0);
  const std::size_t n = f.get_number_of_elements();
  if (n == 0) return;
  hipLaunchKernelGGL((detail::element_kernel<type_t, func_t>), dim3(detail::grid(n, ctx)), dim3(256), 0, ctx.stream(),
                     (const type_t*)f.data(), op, n);
}

template <parallel_for_each_t type, typename func_t, typename graph_t>
std::enable_if_t<type != parallel_for_each_t::element> execute(graph_t& G, func_t op,
                                                               gcuda::multi_context_t& context) {
  using index_t = std::conditional_t<type == parallel_for_each_t::vertex, typename graph_t::vertex_type,
                                     typename graph_t::edge_type>;
  auto& ctx = *context.get_context(0);
  const std::size_t n = (type == parallel_for_each_t::vertex) ? (std::size_t)G.get_number_of_vertices()
                                                              : (std::size_t)G.get_number_of_edges();
  if (n == 0) return;
  hipLaunchKernelGGL((detail::graph_kernel<type, graph_t, func_t, index_t>), dim3(detail::grid(n, ctx)), dim3(256), 0,
                     ctx.stream(), G, op, n);
}

}  // namespace parallel_for
}  // namespace operators
}  // namespace gunrock
