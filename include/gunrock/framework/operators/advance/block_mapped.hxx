// block_mapped.hxx -- a workgroup takes 256 consecutive input slots, stages
// their row starts and block-scanned degrees in LDS, then all threads stride the
// combined edge list (owner found by an 8-probe LDS binary search).
// API parity: include/gunrock/framework/operators/advance/block_mapped.hxx:67-249
// (reference).  Differences: the output base of a workgroup is segments[first
// slot] from the global scan (deterministic order) instead of an atomicAdd on a
// freshly hipMalloc'ed counter per call; the block scan is our wave64 scan, not
// hipcub::BlockScan.
#pragma once

#include <gunrock/framework/operators/advance/helpers.hxx>

namespace gunrock {
namespace operators {
namespace advance {
namespace block_mapped {

template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
__global__ __launch_bounds__(detail::BLOCK) void kernel(graph_t G, operator_t op, const type_t* input,
                                                        std::size_t n, type_t* output, const edge_t* segments) {
  using vertex_t = typename graph_t::vertex_type;
  __shared__ int s_seg[detail::BLOCK + 1];
  __shared__ int s_start[detail::BLOCK];
  __shared__ type_t s_src[detail::BLOCK];
  __shared__ int s_wave[detail::BLOCK / 64 + 1];
  const std::size_t first_slot = (std::size_t)blockIdx.x * detail::BLOCK;
  const std::size_t i = first_slot + threadIdx.x;
  type_t v = gunrock::numeric_limits<type_t>::invalid();
  int start = 0, deg = 0;
  if (i < n) {
    v = input ? input[i] : (type_t)i;
    if (gunrock::util::limits::is_valid(v)) {
      start = (int)G.get_starting_edge((vertex_t)v);
      deg = (int)G.get_number_of_neighbors((vertex_t)v);
    }
  }
  int total;
  const int ex = grx::dev::block_exclusive_sum<detail::BLOCK>(deg, s_wave, &total);
  s_seg[threadIdx.x] = ex;
  s_start[threadIdx.x] = start;
  s_src[threadIdx.x] = v;
  if (threadIdx.x == 0) s_seg[detail::BLOCK] = total;
  __syncthreads();
  type_t* out = nullptr;
  if constexpr (output_type != advance_io_type_t::none) out = output + segments[first_slot];
  detail::expand_window<output_type>(G, op, s_seg, s_start, s_src, detail::BLOCK, 0, total, out);
}

template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
void launch(graph_t& G, operator_t op, const type_t* input, std::size_t n, type_t* output, const edge_t* segments,
            gcuda::standard_context_t& context) {
  if (n == 0) return;
  hipLaunchKernelGGL((kernel<output_type, graph_t, operator_t, type_t, edge_t>),
                     dim3((unsigned)((n + detail::BLOCK - 1) / detail::BLOCK)), dim3(detail::BLOCK), 0,
                     context.stream(), G, op, input, n, output, segments);
}

}  // namespace block_mapped
}  // namespace advance
}  // namespace operators
}  // namespace gunrock
