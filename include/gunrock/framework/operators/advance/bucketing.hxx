// bucketing.hxx -- load balancing chosen PER FRONTIER from a degree histogram.
// The reference declares load_balance_t::bucketing and ships an empty stub
// (include/gunrock/framework/operators/advance/bucketing.hxx:30-35); it also has a
// whole-graph log2 degree histogram helper that no operator uses
// (graph/graph.hxx:393-439).  Here the histogram is taken over the INPUT FRONTIER
// (log2 bins, one wave-aggregated LDS pass + a handful of global atomics) right after
// the degree scan, and the advance kernel is picked from it:
//   * thread_mapped  every degree <= 8 (road-like level): no LDS staging, no search
//   * warp_mapped    mean degree >= 64 and no extreme hub: one wave per row, coalesced rows
//   * block_mapped   moderate skew: a hub costs at most its own workgroup
//   * merge_path     heavy skew (max degree > 64 x mean or > 16 K): hubs must be split
#pragma once

#include <gunrock/framework/operators/advance/helpers.hxx>

namespace gunrock {
namespace operators {
namespace advance {
namespace bucketing {

constexpr int BINS = 32;  // bin b counts degrees in [2^(b-1), 2^b), bin 0 = degree 0

template <typename edge_t>
__global__ __launch_bounds__(256) void histogram_kernel(const edge_t* segments, std::size_t n, unsigned* bins) {
  __shared__ unsigned s_bins[BINS];
  if (threadIdx.x < BINS) s_bins[threadIdx.x] = 0;
  __syncthreads();
  for (std::size_t i = (std::size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (std::size_t)gridDim.x * 256) {
    const unsigned d = (unsigned)(segments[i + 1] - segments[i]);
    const int b = d == 0 ? 0 : 32 - __clz(d);
    // wave-aggregated: one LDS atomic per distinct bin present in the wave
    unsigned long long todo = grx::dev::ballot(true);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const int lb = __builtin_amdgcn_readlane(b, leader);
      const unsigned long long same = grx::dev::ballot(b == lb) & todo;
      if (grx::dev::lane_id() == leader) atomicAdd(&s_bins[lb], (unsigned)__popcll(same));
      todo &= ~same;
    }
  }
  __syncthreads();
  if (threadIdx.x < BINS && s_bins[threadIdx.x]) atomicAdd(&bins[threadIdx.x], s_bins[threadIdx.x]);
}

// Returns the kernel to use for this frontier.  One 128-byte readback.
template <typename edge_t>
load_balance_t select(const edge_t* segments, std::size_t n, std::size_t total_atoms,
                      gcuda::standard_context_t& ctx, unsigned* histogram_out = nullptr) {
  if (n == 0 || total_atoms == 0) return load_balance_t::thread_mapped;
  unsigned* bins = ctx.scratch<unsigned>(2, BINS);
  error::throw_if_exception(hipMemsetAsync(bins, 0, BINS * sizeof(unsigned), ctx.stream()), "histogram reset");
  std::size_t grid = (n + 255) / 256;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL((histogram_kernel<edge_t>), dim3((unsigned)grid), dim3(256), 0, ctx.stream(), segments, n, bins);
  const int* h = ctx.read_back(reinterpret_cast<const int*>(bins), BINS);
  if (histogram_out)
    for (int b = 0; b < BINS; ++b) histogram_out[b] = (unsigned)h[b];
  int top = 0;
  for (int b = 0; b < BINS; ++b)
    if (h[b]) top = b;
  const double mean = (double)total_atoms / (double)n;
  const double max_upper = top == 0 ? 0.0 : (double)(1ull << top);  // degrees in the top bin are < 2^top
  if (top <= 4) return load_balance_t::thread_mapped;                // every degree < 16 and mostly <= 8
  if (max_upper > 16384.0 || max_upper > 64.0 * mean) return load_balance_t::merge_path;
  if (mean >= 64.0) return load_balance_t::warp_mapped;
  return load_balance_t::block_mapped;
}

}  // namespace bucketing
}  // namespace advance
}  // namespace operators
}  // namespace gunrock
