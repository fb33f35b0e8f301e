// scan.hxx -- device-wide exclusive prefix sum (int32) in three launches.
// Used off the per-level hot path (transpose build, generic operators'
// output sizing); replaces thrust::transform_exclusive_scan
// (include/gunrock/framework/operators/advance/helpers.hxx:70-79).
#pragma once

#include <gunrock/hip/wave.hxx>

namespace grx {

constexpr int SCAN_BLOCK = 1024;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// out[i] = sum_{j<i} in[j] for i in [0, n]; out has n + 1 entries.
// `block_sums` needs ceil(n / SCAN_TILE) + 1 ints of scratch.  in == out allowed
// only if out has n + 1 entries and in is read before being overwritten per tile.
static __global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_kernel(const int32_t* in, int64_t n,
                                                                      int32_t* block_sums) {
  __shared__ int s_w[SCAN_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  int acc = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const int64_t i = base + (int64_t)k * SCAN_BLOCK + threadIdx.x;
    if (i < n) acc += in[i];
  }
  acc = dev::wave_sum(acc);
  if (dev::lane_id() == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < SCAN_BLOCK / 64; ++i) t += s_w[i];
    block_sums[blockIdx.x] = t;
  }
}

static __global__ __launch_bounds__(SCAN_BLOCK) void scan_top_kernel(int32_t* block_sums, int nb) {
  __shared__ int s_w[SCAN_BLOCK / 64 + 1];
  int carry = 0;
  for (int base = 0; base < nb; base += SCAN_BLOCK) {
    const int i = base + threadIdx.x;
    const int x = i < nb ? block_sums[i] : 0;
    int tot;
    const int ex = dev::block_exclusive_sum<SCAN_BLOCK>(x, s_w, &tot);
    if (i < nb) block_sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) block_sums[nb] = carry;
}

static __global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const int32_t* in, int64_t n,
                                                                     const int32_t* block_sums,
                                                                     int32_t* out) {
  __shared__ int s_w[SCAN_BLOCK / 64 + 1];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  // thread t owns SCAN_ITEMS consecutive elements so the block scan is over sums
  int v[SCAN_ITEMS];
  int local = 0;
  const int64_t first = base + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (first + k < n) ? in[first + k] : 0;
    local += v[k];
  }
  int tot;
  int ex = dev::block_exclusive_sum<SCAN_BLOCK>(local, s_w, &tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (first + k < n) out[first + k] = ex;
    ex += v[k];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = block_sums[gridDim.x];
}

inline int scan_num_blocks(int64_t n) { return (int)((n + SCAN_TILE - 1) / SCAN_TILE); }

// Enqueue the scan; block_sums must hold scan_num_blocks(n) + 1 ints.  n >= 1.
inline void exclusive_scan_i32(hipStream_t s, const int32_t* in, int64_t n, int32_t* out,
                               int32_t* block_sums) {
  const int nb = scan_num_blocks(n);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n, block_sums);
  hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, block_sums, nb);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n, block_sums, out);
}

}  // namespace grx
