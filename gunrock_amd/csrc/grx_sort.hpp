// grx_sort.hpp -- STABLE least-significant-digit radix sort of (32-bit key, 32-bit value [, second 32-bit value]) on the
// device, for the one-time per-graph preprocessing: transpose (CSC) and the XCD-blocked pull layout.
//
// What it replaces: round 1-3 built those layouts with a counting pass and a fill pass that each did ONE GLOBAL ATOMIC PER
// EDGE (tr_count / tr_fill: 11.4 ms for 69 M edges; xb_count / xb_fill: 124 ms for 182 M edges -- 50-100x off the HBM
// roofline) and left the order inside a column to the arrival order of those atomics.  The reference
// builds its CSC on the HOST with a counting sort (formats/csc.hxx:24-104, graph/conversions/convert.hxx).
//
// Here: passes of <= 9 key bits.  A pass is a per-tile digit histogram (LDS atomics only), one device-wide exclusive scan of
// the (digit, tile) counts, and a scatter in which the position of an element is
//     scanned[digit][tile] + (elements of that digit in earlier waves of the tile) + (rank among the wave's own),
// the last term from ballot matching (the lanes of a wave with the same digit, in lane order) -- no atomic decides an order,
// so the sort is stable and the result is the same on every run and every graph handle.  A wave owns 1024 CONSECUTIVE
// elements of a 4096-element tile and walks them 64 at a time, so all loads are coalesced; stores go out in runs per digit.
// Traffic per pass: 4 B (histogram) + 8 B read + 8 B written per element (12 + 12 with a second value).
#pragma once

#include "grx_common.hpp"
#include <gunrock/hip/scan.hxx>

#include <cstdlib>

namespace grx {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_WAVES = SORT_BLOCK / 64;
constexpr int SORT_STEPS = 16;                               // 64-element steps per wave
constexpr int SORT_TILE = SORT_BLOCK * SORT_STEPS;           // 4096 elements per workgroup
constexpr int SORT_MAX_BITS = 9;
constexpr int SORT_MAX_DIGITS = 1 << SORT_MAX_BITS;
constexpr int SORT_DEFAULT_BITS = 8;

static __global__ __launch_bounds__(SORT_BLOCK) void sort_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                                      int bits, int32_t* hist, int n_tiles) {
  __shared__ int s_h[SORT_MAX_DIGITS];
  const int nd = 1 << bits;
  for (int i = threadIdx.x; i < nd; i += SORT_BLOCK) s_h[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
  const uint32_t mask = (uint32_t)nd - 1u;
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll 4
  for (int k = 0; k < SORT_STEPS; ++k) {
    const int64_t i = base + (int64_t)k * SORT_BLOCK + threadIdx.x;
    const bool valid = i < n;
    const uint32_t d = valid ? (keys[i] >> shift) & mask : 0u;
    // lanes with the same digit add ONCE (the first of them, the group's size): with one LDS atomic per lane a digit that
    // most of a wave shares -- the top digit of a key with few distinct high bits, a hub destination -- is a 64-way
    // serialised atomic per instruction (the XCD-blocked layout's sort: 38 ms for 182 M edges in the first version)
    unsigned long long m = dev::ballot(valid);
    for (int b = 0; b < bits; ++b) {
      const bool one = ((d >> b) & 1u) != 0u;
      const unsigned long long bb = dev::ballot(one);
      m &= one ? bb : ~bb;
    }
    if (valid && (m & lt) == 0ull) atomicAdd(&s_h[d], __popcll(m));
  }
  __syncthreads();
  for (int d = threadIdx.x; d < nd; d += SORT_BLOCK) hist[(size_t)d * (size_t)n_tiles + blockIdx.x] = s_h[d];
}

template <bool V2>
static __global__ __launch_bounds__(SORT_BLOCK) void sort_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                         const uint32_t* __restrict__ vals_in,
                                                                         const uint32_t* __restrict__ vals2_in, int64_t n, int shift,
                                                                         int bits, const int32_t* __restrict__ scanned, int n_tiles,
                                                                         uint32_t* keys_out, uint32_t* vals_out,
                                                                         uint32_t* vals2_out) {
  __shared__ int s_cnt[SORT_WAVES][SORT_MAX_DIGITS];   // per wave and digit: count, then first LOCAL position
  __shared__ int s_delta[SORT_MAX_DIGITS];              // global position of local position i of digit d: s_delta[d] + i
  __shared__ int s_wave[SORT_WAVES + 1];
  __shared__ uint32_t s_key[SORT_TILE], s_val[SORT_TILE], s_val2[V2 ? SORT_TILE : 1];
  const int nd = 1 << bits;
  const uint32_t dmask = (uint32_t)nd - 1u;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < SORT_WAVES * SORT_MAX_DIGITS; i += SORT_BLOCK) (&s_cnt[0][0])[i] = 0;
  const int64_t tile0 = (int64_t)blockIdx.x * SORT_TILE;
  const int64_t base = tile0 + (int64_t)w * (SORT_STEPS * 64) + lane;
  uint32_t key[SORT_STEPS], val[SORT_STEPS], val2[V2 ? SORT_STEPS : 1];
#pragma unroll
  for (int k = 0; k < SORT_STEPS; ++k) {
    const int64_t i = base + (int64_t)k * 64;
    key[k] = i < n ? keys_in[i] : 0u;
    val[k] = i < n ? vals_in[i] : 0u;
    if constexpr (V2) val2[k] = i < n ? vals2_in[i] : 0u;
  }
  __syncthreads();
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int rank[SORT_STEPS];
#pragma unroll
  for (int k = 0; k < SORT_STEPS; ++k) {
    const bool valid = base + (int64_t)k * 64 < n;
    const uint32_t d = (key[k] >> shift) & dmask;
    unsigned long long m = dev::ballot(valid);
    for (int b = 0; b < bits; ++b) {
      const bool one = ((d >> b) & 1u) != 0u;
      const unsigned long long bb = dev::ballot(one);
      m &= one ? bb : ~bb;
    }
    // m: the valid lanes of this step with my digit.  The wave's running count of that digit is read by all of them and
    // advanced by the first (one wave's LDS operations execute in program order)
    const int before = valid ? s_cnt[w][d] : 0;
    rank[k] = before + __popcll(m & lt);
    if (valid && (m & lt) == 0ull) s_cnt[w][d] = before + __popcll(m);
  }
  __syncthreads();
  // counts -> LOCAL positions (the tile sorted by digit, waves in order inside a digit) and the global offset of every digit.
  // Thread t owns the consecutive digits [t * DPT, (t + 1) * DPT): one block scan over the threads' totals.
  {
    const int dpt = (nd + SORT_BLOCK - 1) / SORT_BLOCK;  // 1 or 2
    int tot[2] = {0, 0};
    for (int j = 0; j < dpt; ++j) {
      const int d = tid * dpt + j;
      if (d < nd)
        for (int ww = 0; ww < SORT_WAVES; ++ww) tot[j] += s_cnt[ww][d];
    }
    int block_total;
    int at = dev::block_exclusive_sum<SORT_BLOCK>(tot[0] + tot[1], s_wave, &block_total);
    for (int j = 0; j < dpt; ++j) {
      const int d = tid * dpt + j;
      if (d < nd) {
        s_delta[d] = scanned[(size_t)d * (size_t)n_tiles + blockIdx.x] - at;
        for (int ww = 0; ww < SORT_WAVES; ++ww) {
          const int c = s_cnt[ww][d];
          s_cnt[ww][d] = at;
          at += c;
        }
      }
    }
  }
  __syncthreads();
  // the tile in sorted order through LDS: the global stores below go out in runs (consecutive threads, consecutive
  // addresses inside a digit) instead of 16 scattered 4-byte stores per thread and array
#pragma unroll
  for (int k = 0; k < SORT_STEPS; ++k) {
    if (base + (int64_t)k * 64 < n) {
      const uint32_t d = (key[k] >> shift) & dmask;
      const int pos = s_cnt[w][d] + rank[k];
      s_key[pos] = key[k];
      s_val[pos] = val[k];
      if constexpr (V2) s_val2[pos] = val2[k];
    }
  }
  __syncthreads();
  const int n_here = (int)min((int64_t)SORT_TILE, n - tile0);
#pragma unroll
  for (int k = 0; k < SORT_STEPS; ++k) {
    const int i = k * SORT_BLOCK + tid;
    if (i < n_here) {
      const uint32_t kk = s_key[i];
      const int pos = s_delta[(kk >> shift) & dmask] + i;
      keys_out[pos] = kk;
      vals_out[pos] = s_val[i];
      if constexpr (V2) vals2_out[pos] = s_val2[i];
    }
  }
}

// off[k] = first position whose key (sorted keys >> key_shift) is >= k, for k in [0, n_keys]; off[n_keys] = n.
static __global__ void sort_boundaries_kernel(const uint32_t* __restrict__ sorted_keys, int64_t n, int key_shift, int32_t n_keys,
                                              int32_t* off) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
    // (keys beyond n_keys -- an out-of-range column index in the caller's arrays -- are clamped: nothing is written past off[n_keys])
    const int64_t k_prev = i == 0 ? -1 : min((int64_t)(sorted_keys[i - 1] >> key_shift), (int64_t)n_keys);
    const int64_t k_here = i == n ? (int64_t)n_keys : min((int64_t)(sorted_keys[i] >> key_shift), (int64_t)n_keys);
    for (int64_t k = k_prev + 1; k <= k_here; ++k) off[k] = (int32_t)i;  // (gaps: keys nobody has)
  }
}

// ---- (key, value[, value2]) of every EDGE of a CSR, with its source row, in edge order ----------------------------
// A workgroup takes SORT_TILE consecutive edges.  The rows that begin inside the tile mark their first edge in LDS, an
// inclusive max-scan turns the marks into the row of every edge (the row that is under way where the tile begins comes
// from one binary search per tile), and `emit(e, row, column)` writes the outputs -- coalesced, whatever the degrees are.
// (Round 4, first version: one wave per row -- 2.9 ms for the 69 M edges of the LJ stand-in, whose rows average 14 edges.)
template <typename Emit>
static __global__ __launch_bounds__(SORT_BLOCK) void edge_expand_kernel(const int32_t* __restrict__ ro, const int32_t* __restrict__ ci,
                                                                        int32_t V, int64_t E, Emit emit) {
  __shared__ int s_row[SORT_TILE];
  __shared__ int s_wave[SORT_WAVES + 1];
  __shared__ int s_first[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t e0 = (int64_t)blockIdx.x * SORT_TILE;
  const int n_here = (int)min((int64_t)SORT_TILE, E - e0);
  for (int i = tid; i < SORT_TILE; i += SORT_BLOCK) s_row[i] = -1;
  if (tid < 2) {
    // tid 0: the last row that begins at or before e0 (upper bound - 1); tid 1: the same for the tile's last edge
    const int64_t target = tid == 0 ? e0 : e0 + n_here - 1;
    int lo = 0, hi = V;  // invariant: ro[lo] <= target < ro[hi]  (ro[V] = E > target)
    while (hi - lo > 1) {
      const int mid = lo + (hi - lo) / 2;
      if ((int64_t)ro[mid] <= target) lo = mid; else hi = mid;
    }
    s_first[tid] = lo;
  }
  __syncthreads();
  const int r_first = s_first[0], r_last = s_first[1];
  // rows (r_first, r_last] begin inside the tile (empty rows among them mark nothing)
  for (int r = r_first + 1 + tid; r <= r_last; r += SORT_BLOCK) {
    const int b = ro[r];
    if (ro[r + 1] > b) s_row[(int)((int64_t)b - e0)] = r;
  }
  __syncthreads();
  // inclusive max-scan, thread t owns the SORT_STEPS consecutive positions [t * SORT_STEPS, ...)
  int run = -1, mine[SORT_STEPS];
#pragma unroll
  for (int j = 0; j < SORT_STEPS; ++j) {
    run = max(run, s_row[tid * SORT_STEPS + j]);
    mine[j] = run;
  }
  int inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o, 64);
    if (lane >= o) inc = max(inc, up);
  }
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  int carry = r_first;
  for (int ww = 0; ww < w; ++ww) carry = max(carry, s_wave[ww]);
  const int prev = __shfl_up(inc, 1, 64);
  if (lane > 0) carry = max(carry, prev);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < SORT_STEPS; ++j) s_row[tid * SORT_STEPS + j] = max(mine[j], carry);
  __syncthreads();
#pragma unroll 4
  for (int k = 0; k < SORT_STEPS; ++k) {
    const int i = k * SORT_BLOCK + tid;
    if (i < n_here) emit(e0 + i, s_row[i], ci[e0 + i]);
  }
}

struct sort_buffers {
  uint32_t* keys[2] = {nullptr, nullptr};
  uint32_t* vals[2] = {nullptr, nullptr};
  uint32_t* vals2[2] = {nullptr, nullptr};  // optional second value
  int32_t* hist = nullptr;                  // (1 << SORT_MAX_BITS) * n_tiles + 1 ints, scanned in place
  int32_t* sums = nullptr;                  // block sums of that scan
  int64_t n = 0;
  sort_buffers() = default;
  sort_buffers(const sort_buffers&) = delete;
  sort_buffers& operator=(const sort_buffers&) = delete;
  ~sort_buffers() { release(); }  // whatever was not adopted by the caller (an early error return included)
  hipError_t alloc(int64_t n_, bool second_value) {
    n = n_;
    const size_t bytes = (size_t)std::max<int64_t>(n, 4) * sizeof(uint32_t);
    const int n_tiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    const size_t hn = (size_t)SORT_MAX_DIGITS * (size_t)std::max(1, n_tiles) + 2;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      e = hipMalloc(reinterpret_cast<void**>(&keys[i]), bytes);
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&vals[i]), bytes);
      if (e == hipSuccess && second_value) e = hipMalloc(reinterpret_cast<void**>(&vals2[i]), bytes);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&hist), hn * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&sums), ((size_t)scan_num_blocks((int64_t)hn) + 2) * sizeof(int32_t));
    return e;
  }
  // free everything except the listed pointers (the caller adopts those as its result arrays)
  void release(const void* keep0 = nullptr, const void* keep1 = nullptr, const void* keep2 = nullptr) {
    void* all[] = {keys[0], keys[1], vals[0], vals[1], vals2[0], vals2[1], hist, sums};
    for (void* p : all)
      if (p && p != keep0 && p != keep1 && p != keep2) (void)hipFree(p);
    keys[0] = keys[1] = vals[0] = vals[1] = vals2[0] = vals2[1] = nullptr;
    hist = sums = nullptr;
    n = 0;
  }
};

// Sort the n pairs in buffers [0] by the low `key_bits` bits of the key, stably.  Returns the index (0 | 1) of the buffers
// that hold the result.  Everything is enqueued on `s`; nothing is synchronised.
inline int radix_sort_pairs(hipStream_t s, sort_buffers& b, int key_bits) {
  const int64_t n = b.n;
  if (n <= 0) return 0;
  const int n_tiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
  // digit width: GRX_SORT_BITS (tuning knob, 4..9).  Fewer bits = more passes, but longer runs per digit in the scatter
  int max_bits = SORT_DEFAULT_BITS;
  if (const char* e = getenv("GRX_SORT_BITS")) { const int x = atoi(e); if (x >= 4 && x <= SORT_MAX_BITS) max_bits = x; }
  const int passes = std::max(1, (key_bits + max_bits - 1) / max_bits);
  int cur = 0, shift = 0;
  for (int p = 0; p < passes; ++p) {
    const int bits = (key_bits - shift + (passes - p) - 1) / (passes - p);  // the remaining bits, spread evenly
    const int nd = 1 << bits;
    hipLaunchKernelGGL(sort_hist_kernel, dim3(n_tiles), dim3(SORT_BLOCK), 0, s, b.keys[cur], n, shift, bits, b.hist, n_tiles);
    exclusive_scan_i32(s, b.hist, (int64_t)nd * n_tiles, b.hist, b.sums);
    if (b.vals2[0])
      hipLaunchKernelGGL((sort_scatter_kernel<true>), dim3(n_tiles), dim3(SORT_BLOCK), 0, s, b.keys[cur], b.vals[cur], b.vals2[cur], n,
                         shift, bits, b.hist, n_tiles, b.keys[cur ^ 1], b.vals[cur ^ 1], b.vals2[cur ^ 1]);
    else
      hipLaunchKernelGGL((sort_scatter_kernel<false>), dim3(n_tiles), dim3(SORT_BLOCK), 0, s, b.keys[cur], b.vals[cur], nullptr, n, shift,
                         bits, b.hist, n_tiles, b.keys[cur ^ 1], b.vals[cur ^ 1], nullptr);
    cur ^= 1;
    shift += bits;
  }
  return cur;
}

inline int bits_for(uint64_t n_values) {  // bits needed for keys in [0, n_values)
  int b = 1;
  while (b < 32 && (1ull << b) < n_values) ++b;
  return b;
}

}  // namespace grx
