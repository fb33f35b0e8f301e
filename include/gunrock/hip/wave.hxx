// wave.hxx -- wave64 device primitives for gfx950 (CDNA4).
//
// These replace what the reference takes from hipCUB / rocThrust on the hot
// path (BlockScan in advance/block_mapped.hxx:89,123; transform_exclusive_scan
// in advance/helpers.hxx:70; copy_if in filter/predicated.hxx:30).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace grx {
namespace dev {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Number of set bits of `mask` strictly below this lane.
__device__ __forceinline__ int mask_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ unsigned long long ballot(bool p) {
  return __builtin_amdgcn_ballot_w64(p);
}

// Inclusive prefix sum across the 64 lanes of a wave, on the DPP data path of gfx950: four row_shr steps scan each
// row of 16 lanes, row_bcast:15 / row_bcast:31 carry the row totals across (lanes whose source does not exist, and
// rows masked off, receive the `old` operand, 0).  Six VALU instructions and no address registers -- the
// ds_bpermute form (__shfl_up) it replaces went through the LDS crossbar six times and kept six lane-index VGPRs
// alive in every kernel that scans.
__device__ __forceinline__ int wave_inclusive_sum(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return x;
}

// The same scan for doubles: each DPP step moves the two halves of the operand separately (the DPP path is 32 bits wide) and
// one v_add_f64 follows; lanes without a source receive +0.0.  18 VALU instructions where six __shfl_up of a double were
// twelve trips through the LDS crossbar with their lane arithmetic (round 5: the PageRank pull kernel spent a quarter of its
// instructions there).  The association of the sum is fixed by the schedule: the same bits on every run.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_inclusive_sum_f64(double x) {
  x += dpp_f64<0x111, 0xf>(x);  // row_shr:1
  x += dpp_f64<0x112, 0xf>(x);  // row_shr:2
  x += dpp_f64<0x114, 0xf>(x);  // row_shr:4
  x += dpp_f64<0x118, 0xf>(x);  // row_shr:8
  x += dpp_f64<0x142, 0xa>(x);  // row_bcast:15 -> rows 1, 3
  x += dpp_f64<0x143, 0xc>(x);  // row_bcast:31 -> rows 2, 3
  return x;
}

// Inclusive running maximum of NON-NEGATIVE values across the wave (same DPP schedule; 0 is the identity).
__device__ __forceinline__ int wave_inclusive_max_nonneg(int x) {
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
  return x;
}

// The value of the lane below (lane 0: 0): wave_shr:1.
__device__ __forceinline__ int wave_shift_up1(int x) {
  return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false);
}

// Sum over the 64 lanes, the same value in every lane: the DPP scan above and one v_readlane of its last lane (round 5: the
// butterfly of six __shfl_xor it replaces was six dependent trips through the LDS crossbar, each with its own lane-index
// arithmetic -- 36 VALU instructions and ~0.3 us per call in the inner loops that reduce once per batch).  Wave-wide call.
__device__ __forceinline__ int wave_sum(int x) {
  return __builtin_amdgcn_readlane(wave_inclusive_sum(x), WAVE - 1);
}
// 64-bit sum of non-negative 32-bit terms (degree sums), as two DPP sums of 16-bit halves.
__device__ __forceinline__ long long wave_sum_u32_wide(unsigned x) {
  const int lo = wave_sum((int)(x & 0xffffu)), hi = wave_sum((int)(x >> 16));
  return ((long long)hi << 16) + (long long)lo;
}
// The value lane 0 holds, in every lane (v_readlane; __shfl(x, 0) is a ds_bpermute round trip).  Wave-wide call.
__device__ __forceinline__ int wave_bcast0(int x) { return __builtin_amdgcn_readlane(x, 0); }
// x of the lane whose index differs in bit 0 / bit 1 (quad permutes on the DPP path)
__device__ __forceinline__ int lane_xor1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ int lane_xor2(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false); }

__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, WAVE);
  return x;
}

__device__ __forceinline__ float wave_max_f(float x) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, WAVE));
  return x;
}

// Exclusive prefix sum over a block of NT threads (NT multiple of 64, <= 1024).
// `wave_tot` is LDS scratch of NT/64 + 1 ints.  Returns the exclusive prefix of
// x; *total receives the block sum.  Contains two __syncthreads().
template <int NT>
__device__ __forceinline__ int block_exclusive_sum(int x, int* wave_tot, int* total) {
  constexpr int NW = NT / WAVE;
  const int lane = lane_id();
  const int wid = threadIdx.x / WAVE;
  int inc = wave_inclusive_sum(x);
  if (lane == WAVE - 1) wave_tot[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    int t = wave_tot[i];
    if (i < wid) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - x;
}

// float min through integer atomics: exact for every non-NaN pair (the
// reference uses a CAS loop, cuda/atomic_functions.hxx:34-44).  Returns the
// previous value.
__device__ __forceinline__ float atomic_min_f32(float* addr, float val) {
  if (val >= 0.0f) {
    int old = atomicMin(reinterpret_cast<int*>(addr), __float_as_int(val));
    return __int_as_float(old);
  } else {
    unsigned old = atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(val));
    return __uint_as_float(old);
  }
}

}  // namespace dev
}  // namespace grx
