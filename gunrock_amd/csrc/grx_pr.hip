// grx_pr.hip -- PageRank as an atomics-free pull (transpose SpMV-shaped) iteration.
//
// Reference behaviour reproduced (include/gunrock/algorithms/pr.hxx):
//   reset        :65-93    p = 1/V, iweights[v] = alpha / sum_out_weights(v) (0 if none)
//   loop         :107-152  plast = p; dsum = sum_{iw==0} alpha*p; p = (1-alpha+dsum)/V;
//                          for every edge (s,d,w): p[d] += plast[s]*iw[s]*w   (atomicAdd)
//   is_converged :172-195  iteration > 0 && max|p - plast| < tol
// The reference is edge-parallel: one binary search over row offsets per edge
// (graph/csr.hxx:66-81) plus one float atomicAdd per edge.  Here every vertex
// GATHERS over its in-edges (transpose built once per graph), so there are no
// atomics and no searches; rows are packed into a STATIC partition of <= 2048
// non-zeros per workgroup (long rows are cut into pieces and recombined in a
// fixed order), products are staged in LDS with coalesced index/weight reads.
// The problem is bandwidth bound (~0.17 flop/byte); MFMA has nothing to offer
// at one multiply-add per 12 gathered bytes and is deliberately not used.
#include "grx_engine.hpp"
#include "grx_sort.hpp"

#include <atomic>
#include <thread>
#include <gunrock/hip/scan.hxx>

#include <algorithm>
#include <cstdlib>

namespace grx {

constexpr int PR_BLOCK = 256;
constexpr int PR_NNZ = 2048;   // non-zeros per workgroup
constexpr int PR_LONG = 512;   // rows longer than this are cut into pieces
constexpr int PR_MAXROWS = 2048;

struct pr_args {
  const int32_t* ro;     // CSR (out-edges) for iweights
  const float* w;
  const int32_t* t_ro;   // transpose
  const int32_t* t_ci;
  const float* t_w;
  int32_t V;
  ctrl_t* ctrl;
  float* p;
  float* x;              // plast * iweights
  float* iw;
  float* partial;        // dangling-mass partial sums, one per prepare block
  float* piece_sum;
  const int4* blocks;
  const int32_t* piece;
  const int32_t* longrows;
  int32_t n_blocks, n_long, n_partial;
  float alpha, tol;
  unsigned* err_bits;    // [2] max |p - plast| as ordered uint, by iteration parity
  float* base;           // scalar (1 - alpha + dsum) / V
  // XCD-blocked variant
  const int32_t* xb_ro;
  const int32_t* xb_ci;
  const float* xb_w;
  const int4* xb_blocks;
  const int32_t* xb_piece;
  const int32_t* xb_long;
  int32_t xb_begin[9];
  int32_t n_xb_long;
  float* partial_y;      // NB * V partial sums, block-major
  const int32_t* x_perm; // XCD-blocked variant: position of vertex v's value in x[] (null: v)
  const uint16_t* xb_pos; // XCD-blocked variant: entry i of a row block belongs at position xb_pos[i] of the block (null: i)
  int32_t xb_per_block;   // XCD-blocked variant: x[] positions of source block s begin at s * xb_per_block (hub-first order)
  int32_t hot_n;          // ... and the first hot_n of them are kept in LDS by every workgroup (0: none)
  int32_t xb;             // XCD-blocked variant: source blocks (grx_graph::xb_n: 8 = one per XCD, or 4 / 2 / 1)
  // partitioned run (grx_pr_dist_*): this rank prepares / updates the rows [row_lo, row_hi) only; the dangling mass and
  // the convergence norm are combined over the ranks' {dsum, err} pairs in `gathered` (null: single GPU)
  int32_t row_lo, row_hi;
  int32_t n_ranks;
  unsigned* scal_out;        // this rank's pair, written by pr_dist_pack_kernel
  const unsigned* gathered;  // n_ranks pairs after the all-gather
};

constexpr int XB = 8;  // source blocks at most == XCDs (the number a graph uses: grx_graph::xb_n, pr_args::xb)

// iweights (pr.hxx:78-88): wave per 64 rows; long rows summed cooperatively.
__global__ void pr_iweights_kernel(pr_args a) {
  const int lane = dev::lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r0 = wave * 64; r0 < a.V; r0 += nwaves * 64) {
    const int64_t v = r0 + lane;
    int b = 0, e = 0;
    if (v < a.V) { b = a.ro[v]; e = a.ro[v + 1]; }
    float val = 0.0f;
    if (!a.w) {
      val = (float)(e - b);
    } else {
      const bool is_long = (e - b) >= 128;
      if (!is_long)
        for (int k = b; k < e; ++k) val += a.w[k];
      unsigned long long m = dev::ballot(is_long);
      while (m) {
        const int src_lane = __builtin_ctzll(m);
        m &= m - 1;
        const int bb = __shfl(b, src_lane, 64), ee = __shfl(e, src_lane, 64);
        float part = 0.0f;
        for (int k = bb + lane; k < ee; k += 64) part += a.w[k];
        part = dev::wave_sum_f(part);
        if (lane == src_lane) val = part;
      }
    }
    if (v < a.V) a.iw[v] = (val != 0.0f) ? a.alpha / val : 0.0f;
  }
}

// x = p * iw, dangling partial sums (pr.hxx:121-132).  Fixed block -> range map
// so the partials, and hence dsum, are reproducible.
__global__ __launch_bounds__(PR_BLOCK) void pr_prepare_kernel(pr_args a) {
  __shared__ float s_w[PR_BLOCK / 64];
  if (a.ctrl->done) return;
  const int64_t per = ((int64_t)(a.row_hi - a.row_lo) + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)a.row_lo + (int64_t)blockIdx.x * per, hi = min((int64_t)a.row_hi, lo + per);
  float acc = 0.0f;
  for (int64_t v = lo + threadIdx.x; v < hi; v += PR_BLOCK) {
    const float pv = a.p[v], iwv = a.iw[v];
    a.x[a.x_perm ? a.x_perm[v] : v] = pv * iwv;
    acc += (iwv == 0.0f) ? a.alpha * pv : 0.0f;
  }
  acc = dev::wave_sum_f(acc);
  if (dev::lane_id() == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < PR_BLOCK / 64; ++i) t += s_w[i];
    a.partial[blockIdx.x] = t;
  }
}

// One workgroup: convergence test of the previous iteration, then the scalar
// base term of this one.  `iter` is the index of the loop() about to run.
__global__ __launch_bounds__(PR_BLOCK) void pr_scalar_kernel(pr_args a, int iter) {
  __shared__ float s_w[PR_BLOCK / 64];
  ctrl_t* c = a.ctrl;
  if (c->done) return;
  if (iter > 0) {
    float err = __uint_as_float(a.err_bits[(iter - 1) & 1]);
    if (a.gathered) {  // the norm is the maximum over the ranks (every rank reads the same pairs: same decision)
      err = 0.0f;
      for (int r = 0; r < a.n_ranks; ++r) err = fmaxf(err, __uint_as_float(a.gathered[2 * r + 1]));
    }
    if (err < a.tol) {
      if (threadIdx.x == 0) { c->done = 1; c->pr_iter = iter; c->pr_err = err; }
      return;
    }
  }
  float acc = 0.0f;
  for (int i = threadIdx.x; i < a.n_partial; i += PR_BLOCK) acc += a.partial[i];
  acc = dev::wave_sum_f(acc);
  if (dev::lane_id() == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float dsum = 0.0f;
#pragma unroll
    for (int i = 0; i < PR_BLOCK / 64; ++i) dsum += s_w[i];
    if (a.gathered) {  // partitioned run: the ranks' dangling sums, added in rank order on every rank
      dsum = 0.0f;
      for (int r = 0; r < a.n_ranks; ++r) dsum += __uint_as_float(a.gathered[2 * r]);
    }
    *a.base = (1 - a.alpha + dsum) / a.V;   // (1 - alpha + dsum) / n_vertices, pr.hxx:134
    a.err_bits[iter & 1] = 0u;
    c->pr_dsum = dsum;
    c->pr_iter = iter + 1;
  }
}

// Row sums of a block: an exclusive prefix sum (float64, in LDS) over the block's PR_NNZ products, then
// row sum = prefix[row end] - prefix[row begin].  The first version let one thread add up each row: on a heavy-tailed
// graph most blocks hold one row of a few hundred non-zeros next to rows of ten, and the whole workgroup waited for the
// thread that had it (measured: 0.94 -> 0.77 ms per iteration on the kron stand-in).  float64 makes the
// difference of two prefixes exact to ~1e-16 relative -- closer to the float64 yardstick than the sequential
// float32 sum it replaces -- and the summation order is fixed, so the result is reproducible.
// s_pre: the products on entry, element i at s_pre[pr_slot(i)] -- one double of padding per 16, so that the threads'
// runs of 8 consecutive elements (64 bytes apart) do not all fall into the same two LDS banks; s_wtot: one double per
// wave.  Block-wide.
constexpr int PR_PRE_DOUBLES = PR_NNZ + PR_NNZ / 16 + 2;
__device__ __forceinline__ int pr_slot(int i) { return i + (i >> 4); }
__device__ __forceinline__ void pr_block_prefix(double* s_pre, double* s_wtot) {
  constexpr int PER = PR_NNZ / PR_BLOCK;
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  const int wid = tid >> 6;
  double v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) v[k] = s_pre[pr_slot(tid * PER + k)];
  double run = 0.0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const double x = v[k];
    v[k] = run;  // exclusive inside the thread
    run += x;
  }
  const double inc = dev::wave_inclusive_sum_f64(run);  // inclusive across the wave (DPP: include/gunrock/hip/wave.hxx)
  if (lane == 63) s_wtot[wid] = inc;
  __syncthreads();
  double carry = inc - run;
#pragma unroll
  for (int i = 0; i < PR_BLOCK / 64; ++i)
    if (i < wid) carry += s_wtot[i];
#pragma unroll
  for (int k = 0; k < PER; ++k) s_pre[pr_slot(tid * PER + k)] = carry + v[k];
  if (tid == PR_BLOCK - 1) s_pre[pr_slot(PR_NNZ)] = carry + run;
  __syncthreads();
}

// Partitioned run, before the exchange of iteration `iter`: this rank's {dangling sum of its rows, norm of its rows in
// the previous iteration} -- the pair every rank all-gathers.  Fixed summation order.  <<<1, PR_BLOCK>>>
__global__ __launch_bounds__(PR_BLOCK) void pr_dist_pack_kernel(pr_args a, int iter) {
  __shared__ float s_w[PR_BLOCK / 64];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < a.n_partial; i += PR_BLOCK) acc += a.partial[i];
  acc = dev::wave_sum_f(acc);
  if (dev::lane_id() == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float dsum = 0.0f;
#pragma unroll
    for (int i = 0; i < PR_BLOCK / 64; ++i) dsum += s_w[i];
    a.scal_out[0] = __float_as_uint(dsum);
    a.scal_out[1] = iter > 0 ? a.err_bits[(iter - 1) & 1] : 0x7f7fffffu;
  }
}

__global__ __launch_bounds__(PR_BLOCK) void pr_pull_kernel(pr_args a, int iter) {
  __shared__ double s_pre[PR_PRE_DOUBLES];
  __shared__ double s_wtot[PR_BLOCK / 64];
  __shared__ float s_w[PR_BLOCK / 64];
  if (a.ctrl->done) return;
  const int tid = threadIdx.x;
  const float base = *a.base;
  float err = 0.0f;
  for (int b = blockIdx.x; b < a.n_blocks; b += gridDim.x) {
    const int4 d = a.blocks[b];  // {row0, nrows, e0, e1}
    const int n = d.w - d.z;
    if (d.y > 0) {
      // all index loads first, then all gathers: PR_NNZ / PR_BLOCK independent chains in
      // flight per thread.  The index / weight streams are read once per iteration and are
      // loaded non-temporally so that they do not push x[] (the gathered vector) out of L2.
      constexpr int PER = PR_NNZ / PR_BLOCK;
      int src[PER];
      float wv[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * PR_BLOCK;
        src[k] = i < n ? __builtin_nontemporal_load(&a.t_ci[d.z + i]) : -1;
        wv[k] = (a.t_w && i < n) ? __builtin_nontemporal_load(&a.t_w[d.z + i]) : 1.0f;
      }
      float xv[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) xv[k] = src[k] >= 0 ? a.x[src[k]] : 0.0f;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * PR_BLOCK;
        s_pre[pr_slot(i)] = i < n ? (double)(a.t_w ? xv[k] * wv[k] : xv[k]) : 0.0;
      }
      __syncthreads();
      pr_block_prefix(s_pre, s_wtot);
      for (int r = tid; r < d.y; r += PR_BLOCK) {
        const int row = d.x + r;
        const int s = a.t_ro[row] - d.z, t = a.t_ro[row + 1] - d.z;
        const float acc = (float)(s_pre[pr_slot(t)] - s_pre[pr_slot(s)]);
        const float np = base + acc;
        err = fmaxf(err, fabsf(np - a.p[row]));
        a.p[row] = np;
      }
      __syncthreads();
    } else {
      // piece of a long row: fixed-shape tree => reproducible partial
      constexpr int PER = PR_NNZ / PR_BLOCK;
      int src[PER];
      float wv[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * PR_BLOCK;
        src[k] = i < n ? __builtin_nontemporal_load(&a.t_ci[d.z + i]) : -1;
        wv[k] = (a.t_w && i < n) ? __builtin_nontemporal_load(&a.t_w[d.z + i]) : 1.0f;
      }
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < PER; ++k) {  // fixed order => reproducible partial
        const float xv = src[k] >= 0 ? a.x[src[k]] : 0.0f;
        acc += a.t_w ? xv * wv[k] : xv;
      }
      acc = dev::wave_sum_f(acc);
      if (dev::lane_id() == 0) s_w[tid >> 6] = acc;
      __syncthreads();
      if (tid == 0) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < PR_BLOCK / 64; ++i) t += s_w[i];
        a.piece_sum[a.piece[b]] = t;
      }
      __syncthreads();
    }
  }
  err = dev::wave_max_f(err);
  if (dev::lane_id() == 0 && err > 0.0f) atomicMax(&a.err_bits[iter & 1], __float_as_uint(err));
}

__global__ void pr_long_kernel(pr_args a, int iter) {
  if (a.ctrl->done) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_long) return;
  const int row = a.longrows[3 * i], first = a.longrows[3 * i + 1], np = a.longrows[3 * i + 2];
  float acc = 0.0f;
  for (int k = 0; k < np; ++k) acc += a.piece_sum[first + k];
  const float v = *a.base + acc;
  const float err = fabsf(v - a.p[row]);
  a.p[row] = v;
  if (err > 0.0f) atomicMax(&a.err_bits[iter & 1], __float_as_uint(err));
}

// ---- XCD-blocked pull -------------------------------------------------------------
// The 8 XCDs of an MI355X have private 4 MB L2s; workgroup b is dispatched to XCD b % 8
// (observed placement; used for speed only).  In-edges are bucketed by SOURCE block s
// (V/8 consecutive vertices), and workgroups with b % 8 == s process bucket s only, so the
// slice of x[] they gather from (V/8 floats: 1 MB for 2 M vertices) stays resident in
// THEIR L2 instead of 8 MB of x[] thrashing through every L2.  Each (s, row) gets a partial
// sum; pr_combine_kernel adds the 8 partials of a row in fixed order.
//
// THE HOT HEAD OF THE SLICE IN LDS (round 5).  What bounds this kernel, from the counters (profiles/r5_c18_pr_counters.md): the
// CU's L1 is busy 92 % of the launch and stalled on pending misses 61 % of it -- 244 k L2 requests per CU and launch at 228
// cycles each, against the few dozen misses an L1 keeps in flight.  Not occupancy (4 workgroups per CU run within 3 % of 8),
// not address coalescing (row blocks sorted by source changed nothing on this graph).  The sources are ranked hub-first, so the
// first hot_n positions of the slice receive most of the gathers: every workgroup copies them into LDS once per launch and
// gathers from there; only the cold tail goes through the L1.  (Round 2 measured an LDS copy SLOWER -- on a kernel that still
// depended on its resident workgroups.)
template <bool HOT>
__global__ __launch_bounds__(PR_BLOCK) void pr_pull_xcd_kernel(pr_args a) {
  __shared__ double s_pre[PR_PRE_DOUBLES];
  __shared__ double s_wtot[PR_BLOCK / 64];
  __shared__ float s_w[PR_BLOCK / 64];
  extern __shared__ float s_hot[];  // hot_n floats (dynamic)
  if (a.ctrl->done) return;
  const int tid = threadIdx.x;
  const int s = blockIdx.x % a.xb;
  const int lane_b = blockIdx.x / a.xb, stride_b = gridDim.x / a.xb;
  const int32_t* ro = a.xb_ro + (size_t)s * ((size_t)a.V + 1);
  float* y = a.partial_y + (size_t)s * (size_t)a.V;
  constexpr int PER = PR_NNZ / PR_BLOCK;
  const int hot_n = HOT ? a.hot_n : 0;
  const int hot_base = s * a.xb_per_block;
  if constexpr (HOT) {
    for (int i = tid; i < hot_n; i += PR_BLOCK) s_hot[i] = a.x[hot_base + i];
    __syncthreads();
  }
  // x[src], from LDS when src lies in the hot head of this slice (src < 0: a lane past the end of the block)
  auto gather = [&](int src) -> float {
    if constexpr (!HOT) return src >= 0 ? a.x[src] : 0.0f;
    const unsigned loc = (unsigned)(src - hot_base);
    float v = 0.0f;
    if (src >= 0 && loc >= (unsigned)hot_n) v = a.x[src];          // (exec-masked: hot lanes issue no request)
    const float h = s_hot[loc < (unsigned)hot_n ? loc : 0u];        // (unconditional LDS read from a clamped index)
    return (src >= 0 && loc < (unsigned)hot_n) ? h : v;
  };
  for (int b = a.xb_begin[s] + lane_b; b < a.xb_begin[s + 1]; b += stride_b) {
    const int4 d = a.xb_blocks[b];  // {row0, nrows, e0, e1}; nrows == 0 => piece of a long row
    const int n = d.w - d.z;
    int src[PER];
    float wv[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * PR_BLOCK;
      src[k] = i < n ? __builtin_nontemporal_load(&a.xb_ci[d.z + i]) : -1;
      wv[k] = (a.xb_w && i < n) ? __builtin_nontemporal_load(&a.xb_w[d.z + i]) : 1.0f;
    }
    if (d.y > 0) {
      // The entries of a row block are stored SORTED BY SOURCE (round 5, build_pr_xcd_layout): lanes on consecutive entries
      // gather from ascending addresses, and the sources most entries point at -- the hub-first head of the slice -- sit
      // in a few lines that neighbouring lanes now share.  Every entry carries the position it had in row order (16 bits);
      // its product goes THERE, so the prefix sums and every row sum see exactly the order they always saw.
      int pos[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * PR_BLOCK;
        pos[k] = (a.xb_pos && i < n) ? (int)__builtin_nontemporal_load(&a.xb_pos[d.z + i]) : i;
      }
      float xv[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) xv[k] = gather(src[k]);
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * PR_BLOCK;
        s_pre[pr_slot(pos[k])] = i < n ? (double)(a.xb_w ? xv[k] * wv[k] : xv[k]) : 0.0;
      }
      __syncthreads();
      pr_block_prefix(s_pre, s_wtot);
      for (int r = tid; r < d.y; r += PR_BLOCK) {
        const int row = d.x + r;
        const int lo = ro[row] - d.z, hi = ro[row + 1] - d.z;
        y[row] = (float)(s_pre[pr_slot(hi)] - s_pre[pr_slot(lo)]);
      }
      __syncthreads();
    } else {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const float xv = gather(src[k]);
        acc += a.xb_w ? xv * wv[k] : xv;
      }
      acc = dev::wave_sum_f(acc);
      if (dev::lane_id() == 0) s_w[tid >> 6] = acc;
      __syncthreads();
      if (tid == 0) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < PR_BLOCK / 64; ++i) t += s_w[i];
        a.piece_sum[a.xb_piece[b]] = t;
      }
      __syncthreads();
    }
  }
}

__global__ void pr_long_xcd_kernel(pr_args a) {
  if (a.ctrl->done) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_xb_long) return;
  const int idx = a.xb_long[3 * i], first = a.xb_long[3 * i + 1], np = a.xb_long[3 * i + 2];
  float acc = 0.0f;
  for (int k = 0; k < np; ++k) acc += a.piece_sum[first + k];
  a.partial_y[idx] = acc;  // idx = s * V + row
}

// The 8 partial sums of a row, in fixed order -> the new rank; and, in the same pass, what pr_prepare_kernel would do
// for the NEXT iteration (x = p * iweights, this block's share of the dangling mass): one sweep over p and iw less per
// iteration.  The block -> vertices map is fixed by the grid, so the partials (and dsum) stay reproducible.
__global__ __launch_bounds__(256) void pr_combine_kernel(pr_args a, int iter) {
  __shared__ float s_w[256 / 64];
  if (a.ctrl->done) return;
  const float base = *a.base;
  float err = 0.0f, dang = 0.0f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < a.V; v += (int64_t)gridDim.x * 256) {
    float acc = 0.0f;
    if (a.xb == XB) {
#pragma unroll
      for (int s = 0; s < XB; ++s) acc += a.partial_y[(size_t)s * (size_t)a.V + v];
    } else {
      for (int s = 0; s < a.xb; ++s) acc += a.partial_y[(size_t)s * (size_t)a.V + v];
    }
    const float np = base + acc;
    err = fmaxf(err, fabsf(np - a.p[v]));
    a.p[v] = np;
    const float iwv = a.iw[v];
    a.x[a.x_perm ? a.x_perm[v] : v] = np * iwv;
    dang += (iwv == 0.0f) ? a.alpha * np : 0.0f;
  }
  err = dev::wave_max_f(err);
  if (dev::lane_id() == 0 && err > 0.0f) atomicMax(&a.err_bits[iter & 1], __float_as_uint(err));
  dang = dev::wave_sum_f(dang);
  if (dev::lane_id() == 0) s_w[threadIdx.x >> 6] = dang;
  __syncthreads();
  if (threadIdx.x == 0) a.partial[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// The XCD-blocked layout is the edge list sorted by (source block, destination): key = block * (V + 1) + destination,
// value = the (relabelled) source [, weight].  `perm` (may be null) relabels the SOURCES: in-edges are bucketed by, and
// store, perm[u].  (edge_expand_kernel, grx_sort.hpp, supplies the source row of every edge.)
struct xb_emit {
  const float* w;
  const int32_t* perm;
  int32_t V, per_block;
  uint32_t* keys;
  uint32_t* vals;
  uint32_t* vals2;
  __device__ __forceinline__ void operator()(int64_t e, int row, int col) const {
    const int32_t pu = perm ? perm[row] : row;
    keys[e] = (uint32_t)(pu / per_block) * ((uint32_t)V + 1u) + (uint32_t)col;
    vals[e] = (uint32_t)pu;
    if (vals2) vals2[e] = __float_as_uint(w[e]);
  }
};

// hub-first ranking on the device: vertices sorted by out-degree, descending, ties in id order (a stable sort by the
// complemented degree); rank r goes to block r % XB, position r / XB
__global__ void xb_rank_keys_kernel(const int32_t* __restrict__ ro, int32_t V, uint32_t* keys, uint32_t* vals) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    keys[v] = ~(uint32_t)(ro[v + 1] - ro[v]);
    vals[v] = (uint32_t)v;
  }
}
__global__ void xb_rank_perm_kernel(const uint32_t* __restrict__ order, int32_t V, int32_t per_block, int32_t xb, int32_t* perm) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < V; r += (int64_t)gridDim.x * blockDim.x)
    perm[order[r]] = (int32_t)((r % xb) * per_block + r / xb);
}

__global__ void pr_init_kernel(pr_args a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    ctrl_t* c = a.ctrl;
    c->done = 0;
    c->level = 0;
    c->pr_iter = 0;
    c->pr_err = 0.0f;
    a.err_bits[0] = 0u;
    a.err_bits[1] = 0u;
  }
}

// Static pull partition, built once per graph on the host from the transpose
// offsets: consecutive short rows are packed up to PR_NNZ non-zeros (and
// PR_MAXROWS rows); rows longer than PR_LONG become pieces of PR_NNZ.
struct pr_partition {
  void* blocks = nullptr;
  int32_t* piece = nullptr;
  int32_t* longrows = nullptr;
  int32_t n_blocks = 0, n_pieces = 0, n_long = 0;
};
// The packing of rows [lo, hi) of the matrix whose (host) row offsets are `ro`, as host vectors.  `row_id_base` is added to
// the row ids recorded for long rows (the XCD-blocked lists number them block-major).
struct pr_pack {
  std::vector<int4> blocks;
  std::vector<int32_t> piece, longrows;  // piece: -1 or the piece number LOCAL to this pack; longrows: {row, first piece (local), n}
  int32_t n_pieces = 0;
};
static void pack_rows(const int32_t* ro, int32_t lo, int32_t hi, int64_t row_id_base, pr_pack& out) {
  int32_t row0 = lo;
  auto flush = [&](int32_t row_end) {
    if (row_end > row0) {
      out.blocks.push_back(make_int4(row0, row_end - row0, ro[row0], ro[row_end]));
      out.piece.push_back(-1);
    }
    row0 = row_end;
  };
  for (int32_t v = lo; v < hi; ++v) {
    const int32_t deg = ro[v + 1] - ro[v];
    if (deg > PR_LONG) {
      flush(v);
      out.longrows.push_back((int32_t)(row_id_base + v));
      out.longrows.push_back(out.n_pieces);
      int32_t np = 0;
      for (int32_t e = ro[v]; e < ro[v + 1]; e += PR_NNZ) {
        out.blocks.push_back(make_int4(v, 0, e, std::min(ro[v + 1], e + PR_NNZ)));
        out.piece.push_back(out.n_pieces++);
        ++np;
      }
      out.longrows.push_back(np);
      row0 = v + 1;
      continue;
    }
    if (ro[v + 1] - ro[row0] > PR_NNZ || v - row0 >= PR_MAXROWS) flush(v);
  }
  flush(hi);
}
// The same packing over FIXED chunks of the row range, the chunks packed by host threads in parallel and concatenated in
// order.  The chunk boundaries depend on nothing but (lo, hi): the partition -- and with it every fp32 row sum -- is the same
// on every machine and every handle.  (One thread walked 16.8 M rows of the kron stand-in's eight lists in ~50 ms.)
constexpr int32_t PR_PACK_CHUNK = 1 << 16;
static void pack_rows_parallel(const int32_t* ro, int32_t lo, int32_t hi, int64_t row_id_base, pr_pack& out) {
  const int n_chunks = std::max(1, (hi - lo + PR_PACK_CHUNK - 1) / PR_PACK_CHUNK);
  std::vector<pr_pack> parts((size_t)n_chunks);
  std::atomic<int> next{0};
  const int nt = std::max(1, std::min(n_chunks, std::min(32, (int)std::thread::hardware_concurrency())));
  auto work = [&] {
    for (int c; (c = next.fetch_add(1)) < n_chunks;)
      pack_rows(ro, lo + c * PR_PACK_CHUNK, std::min(hi, lo + (c + 1) * PR_PACK_CHUNK), row_id_base, parts[(size_t)c]);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
  for (auto& p : parts) {
    const int32_t piece0 = out.n_pieces;
    out.blocks.insert(out.blocks.end(), p.blocks.begin(), p.blocks.end());
    for (int32_t x : p.piece) out.piece.push_back(x < 0 ? -1 : x + piece0);
    for (size_t i = 0; i < p.longrows.size(); i += 3) {
      out.longrows.push_back(p.longrows[i]);
      out.longrows.push_back(p.longrows[i + 1] + piece0);
      out.longrows.push_back(p.longrows[i + 2]);
    }
    out.n_pieces += p.n_pieces;
  }
}
static grx_status_t build_pr_partition_rows(const std::vector<int32_t>& ro, int32_t lo, int32_t hi, pr_partition* out) {
  pr_pack pk;
  pack_rows_parallel(ro.data(), lo, hi, 0, pk);
  const std::vector<int4>& blocks = pk.blocks;
  const std::vector<int32_t>&piece = pk.piece, &longrows = pk.longrows;
  out->n_blocks = (int32_t)blocks.size();
  out->n_pieces = pk.n_pieces;
  out->n_long = (int32_t)(longrows.size() / 3);
  GRX_HIP(hipMalloc(&out->blocks, std::max<size_t>(1, blocks.size()) * sizeof(int4)));
  GRX_HIP(hipMalloc(reinterpret_cast<void**>(&out->piece), std::max<size_t>(1, piece.size()) * sizeof(int32_t)));
  GRX_HIP(hipMalloc(reinterpret_cast<void**>(&out->longrows), std::max<size_t>(1, longrows.size()) * sizeof(int32_t)));
  if (!blocks.empty()) {
    GRX_HIP(hipMemcpy(out->blocks, blocks.data(), blocks.size() * sizeof(int4), hipMemcpyHostToDevice));
    GRX_HIP(hipMemcpy(out->piece, piece.data(), piece.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (!longrows.empty())
    GRX_HIP(hipMemcpy(out->longrows, longrows.data(), longrows.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return GRX_SUCCESS;
}
// ---- the same packing ON THE DEVICE (round 4): one thread per chunk of PR_DEV_CHUNK rows walks its rows greedily; a
// counting pass, three tiny scans, a writing pass.  The single-GPU engine builds both of its partitions this way -- no
// host copy of the offsets (67 MB for the eight lists of the kron stand-in), no host loop over 16.8 M rows.  The chunk
// boundaries depend on (rows, chunk size) only, so the partition is the same on every machine and handle.
constexpr int32_t PR_DEV_CHUNK = 4096;
struct pr_pack_args {
  const int32_t* ro;      // offsets of list 0; list l starts at ro + l * list_stride
  int64_t list_stride;
  int32_t n_rows, n_lists, chunks_per_list;
  int32_t* cnt;           // [3][n_chunks + 1]: blocks, pieces, long rows per chunk (pass 0), then their exclusive scans
  int4* blocks;
  int32_t* piece;
  int32_t* longrows;
};
template <int PASS>
__global__ void pr_pack_kernel(pr_pack_args p) {
  const int n_chunks = p.n_lists * p.chunks_per_list;
  const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (c >= n_chunks) return;
  const int list = c / p.chunks_per_list, cc = c % p.chunks_per_list;
  const int32_t* ro = p.ro + (int64_t)list * p.list_stride;
  const int32_t lo = cc * PR_DEV_CHUNK, hi = min(p.n_rows, lo + PR_DEV_CHUNK);
  const int64_t row_id_base = (int64_t)list * (int64_t)p.n_rows;
  int32_t nb = 0, np = 0, nl = 0;
  if (PASS == 1) { nb = p.cnt[c]; np = p.cnt[(n_chunks + 1) + c]; nl = p.cnt[2 * (n_chunks + 1) + c]; }
  int32_t row0 = lo, e_row0 = ro[lo], e_v = e_row0;
  for (int32_t v = lo; v < hi; ++v) {
    const int32_t e_next = ro[v + 1];
    const int32_t deg = e_next - e_v;
    auto flush = [&](int32_t row_end, int32_t e_end) {
      if (row_end > row0) {
        if (PASS == 1) { p.blocks[nb] = make_int4(row0, row_end - row0, e_row0, e_end); p.piece[nb] = -1; }
        ++nb;
      }
      row0 = row_end;
      e_row0 = e_end;
    };
    if (deg > PR_LONG) {
      flush(v, e_v);
      const int32_t pieces = (deg + PR_NNZ - 1) / PR_NNZ;
      if (PASS == 1) {
        p.longrows[3 * nl] = (int32_t)(row_id_base + v);
        p.longrows[3 * nl + 1] = np;
        p.longrows[3 * nl + 2] = pieces;
        for (int32_t k = 0; k < pieces; ++k) {
          p.blocks[nb + k] = make_int4(v, 0, e_v + k * PR_NNZ, min(e_next, e_v + (k + 1) * PR_NNZ));
          p.piece[nb + k] = np + k;
        }
      }
      nb += pieces;
      np += pieces;
      ++nl;
      row0 = v + 1;
      e_row0 = e_next;
    } else if (e_next - e_row0 > PR_NNZ || v - row0 >= PR_MAXROWS) {
      flush(v, e_v);
    }
    e_v = e_next;
  }
  if (hi > row0) {
    if (PASS == 1) { p.blocks[nb] = make_int4(row0, hi - row0, e_row0, e_v); p.piece[nb] = -1; }
    ++nb;
  }
  if (PASS == 0) {
    p.cnt[c] = nb;
    p.cnt[(n_chunks + 1) + c] = np;
    p.cnt[2 * (n_chunks + 1) + c] = nl;
  }
}

// d_ro: n_lists lists of n_rows + 1 offsets, list_stride apart.  list_begin (n_lists + 1 ints, may be null): first block of
// every list.
static grx_status_t build_pr_partition_device(grx_context_t ctx, const int32_t* d_ro, int64_t list_stride, int32_t n_rows, int n_lists,
                                              pr_partition* out, int32_t* list_begin) {
  hipStream_t s = ctx->stream;
  const int cpl = std::max(1, (n_rows + PR_DEV_CHUNK - 1) / PR_DEV_CHUNK), n_chunks = cpl * n_lists;
  dev_scratch cnt_buf, sums_buf, blocks_buf, piece_buf, long_buf;  // (an error return frees what has been allocated so far)
  GRX_HIP(cnt_buf.alloc((size_t)3 * (n_chunks + 1) * sizeof(int32_t)));
  GRX_HIP(sums_buf.alloc(((size_t)scan_num_blocks(n_chunks) + 2) * sizeof(int32_t)));
  int32_t* cnt = cnt_buf.as<int32_t>();
  int32_t* sums = sums_buf.as<int32_t>();
  pr_pack_args p{};
  p.ro = d_ro; p.list_stride = list_stride; p.n_rows = n_rows; p.n_lists = n_lists; p.chunks_per_list = cpl; p.cnt = cnt;
  const dim3 grid((unsigned)((n_chunks + 63) / 64)), block(64);
  hipLaunchKernelGGL(pr_pack_kernel<0>, grid, block, 0, s, p);
  for (int k = 0; k < 3; ++k) exclusive_scan_i32(s, cnt + (size_t)k * (n_chunks + 1), n_chunks, cnt + (size_t)k * (n_chunks + 1), sums);
  std::vector<int32_t> h((size_t)3 * (n_chunks + 1));
  GRX_HIP(hipMemcpyAsync(h.data(), cnt, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  GRX_HIP(hipStreamSynchronize(s));
  out->n_blocks = h[(size_t)n_chunks];
  out->n_pieces = h[(size_t)(n_chunks + 1) + n_chunks];
  out->n_long = h[(size_t)2 * (n_chunks + 1) + n_chunks];
  if (list_begin)
    for (int l = 0; l <= n_lists; ++l) list_begin[l] = l < n_lists ? h[(size_t)l * cpl] : out->n_blocks;
  GRX_HIP(blocks_buf.alloc(std::max<size_t>(1, (size_t)out->n_blocks) * sizeof(int4)));
  GRX_HIP(piece_buf.alloc(std::max<size_t>(1, (size_t)out->n_blocks) * sizeof(int32_t)));
  GRX_HIP(long_buf.alloc(std::max<size_t>(1, (size_t)out->n_long * 3) * sizeof(int32_t)));
  p.blocks = blocks_buf.as<int4>();
  p.piece = piece_buf.as<int32_t>();
  p.longrows = long_buf.as<int32_t>();
  hipLaunchKernelGGL(pr_pack_kernel<1>, grid, block, 0, s, p);
  GRX_HIP(hipStreamSynchronize(s));
  GRX_HIP(hipGetLastError());
  out->blocks = blocks_buf.release();
  out->piece = reinterpret_cast<int32_t*>(piece_buf.release());
  out->longrows = reinterpret_cast<int32_t*>(long_buf.release());
  return GRX_SUCCESS;
}

static grx_status_t build_pr_partition(grx_graph_t g) {
  std::lock_guard<std::recursive_mutex> lk(g->prep_mu);
  if (g->pr_blocks) return GRX_SUCCESS;
  prep_timer tm("pagerank: static partition (device)", g->ctx->stream);
  pr_partition pt;
  grx_status_t st = build_pr_partition_device(g->ctx, g->t_ro, 0, g->V, 1, &pt, nullptr);
  if (st != GRX_SUCCESS) return st;
  g->pr_piece = pt.piece;
  g->pr_long = pt.longrows;
  g->n_pr_blocks = pt.n_blocks;
  g->n_pr_pieces = pt.n_pieces;
  g->n_pr_long = pt.n_long;
  g->pr_blocks = pt.blocks;  // (last: it is what says "built")
  return GRX_SUCCESS;
}

// Sort the entries of every ROW block of the XCD-blocked layout by source (= by the address they gather from), remembering
// where each came from (xb_pos).  One workgroup per block, bitonic network over (source << 16 | position) in LDS: the keys are
// distinct, so the result does not depend on the schedule.  Pieces of long rows (nrows == 0) keep their order -- they are one
// row's ascending sources already, and their sums are taken in storage order.  Once per graph.  <<<n_blocks, 256>>>
__global__ __launch_bounds__(256) void xb_sort_blocks_kernel(const int4* __restrict__ blocks, int n_blocks, int32_t* ci, float* w,
                                                             uint16_t* pos) {
  __shared__ unsigned long long key[PR_NNZ];
  __shared__ float wv[PR_NNZ];
  const int tid = threadIdx.x;
  for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
    const int4 d = blocks[b];
    const int n = d.w - d.z;
    if (d.y == 0) {
      for (int i = tid; i < n; i += 256) pos[d.z + i] = (uint16_t)i;
      continue;
    }
    for (int i = tid; i < PR_NNZ; i += 256) {
      key[i] = i < n ? (((unsigned long long)(unsigned)ci[d.z + i] << 16) | (unsigned long long)i) : ~0ull;
      if (w) wv[i] = i < n ? w[d.z + i] : 0.0f;
    }
    __syncthreads();
    for (int k = 2; k <= PR_NNZ; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < PR_NNZ; i += 256) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long x = key[i], y = key[ixj];
            const bool up = (i & k) == 0;
            if ((x > y) == up) { key[i] = y; key[ixj] = x; }
          }
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < n; i += 256) {
      const unsigned long long kv = key[i];
      const int from = (int)(kv & 0xffffull);
      ci[d.z + i] = (int32_t)(kv >> 16);
      pos[d.z + i] = (uint16_t)from;
      if (w) w[d.z + i] = wv[from];
    }
    __syncthreads();
  }
}

static grx_status_t build_pr_xcd_layout(grx_context_t ctx, grx_graph_t g) {
  std::lock_guard<std::recursive_mutex> lk(g->prep_mu);  // (has_xb is published last, below)
  if (g->has_xb) return GRX_SUCCESS;
  prep_timer tm("pagerank: XCD-blocked layout (rank + sort + partition)", ctx->stream);
  const int32_t V = g->V;
  const int64_t E = g->E;
  // SOURCE BLOCKS (round 5): 8 -- one per XCD, every slice of the gathered vector resident in one L2 -- costs 8 V row visits
  // and 8 V partial sums per iteration; where a row has few entries per block those rival the E gathers.  Measured (profiles/
  // r5_c25_pr_source_blocks.txt), ms per iteration with 8 / 4 / 2 / 1 blocks: LJ stand-in (14 edges per vertex) 0.452 / 0.390 /
  // 0.433 / 0.450, kron stand-in (87) 0.610 / 0.618 / 0.700 / 0.869 -- four blocks below 32 edges per vertex, else eight.
  // GRX_PR_XB = 1 | 2 | 4 | 8: that many.
  int xbn = (E < 32ll * V) ? 4 : XB;
  if (const char* e = getenv("GRX_PR_XB")) {
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8) xbn = v;
  }
  g->xb_n = xbn;
  const size_t n_off = (size_t)xbn * ((size_t)V + 1);
  const int32_t per_block = (V + xbn - 1) / xbn;
  hipStream_t s = ctx->stream;
  // HUB-FIRST RELABELLING OF THE SOURCES.  The gathers of a bucket go to one slice of x[] (V/8
  // floats); on a scale-free graph most of them go to a few thousand hub vertices, which the
  // Graph500 permutation scatters one per cache line.  Sources are ranked by out-degree (= how
  // often they are gathered) and rank r goes to block r % 8, position r / 8: every block gets an
  // equal share of the hubs, packed at the START of its slice, 32 hub values per 128-byte line --
  // a hot set of a few KB that stays in the CU's 32 KB L1 instead of costing an L2 request per
  // gather.  Only the position of x values changes; rows, and the result vector, keep their ids.
  // Measured on the kron stand-in: 1.19 -> 1.03 ms per iteration.  Going further and copying the
  // first 4096 / 8192 values of the slice into LDS per workgroup (62 % / 73 % of all gathers) was
  // SLOWER (1.30 / 1.76 ms): the LDS it takes costs more resident workgroups than the L2
  // requests it saves -- the kernel lives on concurrency, not on request rate.
  if (!getenv("GRX_PR_NOPERM")) {
    prep_timer t0("  xcd layout: hub-first ranking (device sort)", s);
    sort_buffers rb;
    if (rb.alloc(V, false) != hipSuccess) {
      (void)hipGetLastError();
      return fail(GRX_ERROR_OUT_OF_MEMORY, "pagerank: scratch for the hub-first ranking");
    }
    if (!g->xb_perm) GRX_HIP(hipMalloc(reinterpret_cast<void**>(&g->xb_perm), (size_t)V * sizeof(int32_t)));  // (kept by a retry)
    hipLaunchKernelGGL(xb_rank_keys_kernel, dim3(1024), dim3(256), 0, s, g->ro, V, rb.keys[0], rb.vals[0]);
    const int rr = radix_sort_pairs(s, rb, 32);
    hipLaunchKernelGGL(xb_rank_perm_kernel, dim3(1024), dim3(256), 0, s, rb.vals[rr], V, per_block, xbn, g->xb_perm);
    GRX_HIP(hipStreamSynchronize(s));
    rb.release();
  }
  // (an earlier attempt that failed behind this point may have left a buffer sized for ANOTHER block count -- GRX_PR_XB and E / V are
  // read afresh on every attempt: never reuse it)
  if (g->xb_ro) { (void)hipFree(g->xb_ro); g->xb_ro = nullptr; }
  GRX_HIP(hipMalloc(reinterpret_cast<void**>(&g->xb_ro), (n_off + 2) * sizeof(int32_t)));
  const bool unit = graph_unit_weights(g);  // no weight stream needed (graph_weight_stats ran)
  // stable radix sort of the edges by (source block, destination) (grx_sort.hpp): the round 1-3 version counted and filled
  // with one global atomic per edge -- 124 ms for the 182 M edges of the kron stand-in
  sort_buffers sb;
  if ((uint64_t)n_off >= (1ull << 31) || sb.alloc(E, !unit) != hipSuccess) {
    (void)hipGetLastError();
    return fail(GRX_ERROR_OUT_OF_MEMORY, "pagerank: scratch for the XCD-blocked layout");
  }
  int res;
  {
    prep_timer t1("  xcd layout: expand + radix sort", s);
    const xb_emit em{unit ? nullptr : g->w, g->xb_perm, V, per_block, sb.keys[0], sb.vals[0], sb.vals2[0]};
    hipLaunchKernelGGL((edge_expand_kernel<xb_emit>), dim3((unsigned)((E + SORT_TILE - 1) / SORT_TILE)), dim3(SORT_BLOCK), 0, s, g->ro,
                       g->ci, V, E, em);
    res = radix_sort_pairs(s, sb, bits_for((uint64_t)n_off));
  }
  hipLaunchKernelGGL(sort_boundaries_kernel, dim3(2048), dim3(256), 0, s, sb.keys[res], E, 0, (int32_t)n_off, g->xb_ro);
  GRX_HIP(hipStreamSynchronize(s));
  GRX_HIP(hipGetLastError());
  if (g->xb_ci) (void)hipFree(g->xb_ci);  // (left by an attempt that failed behind this point)
  if (g->xb_w) (void)hipFree(g->xb_w);
  g->xb_w = nullptr;
  g->xb_ci = reinterpret_cast<int32_t*>(sb.vals[res]);
  if (!unit) g->xb_w = reinterpret_cast<float*>(sb.vals2[res]);
  sb.release(g->xb_ci, g->xb_w);
  // static partition, one list per source block (same packing rule as the plain layout), built on the device
  pr_partition pt;
  prep_timer t2("  xcd layout: partition (device)", s);
  grx_status_t pst = build_pr_partition_device(ctx, g->xb_ro, (int64_t)V + 1, V, xbn, &pt, g->xb_begin);
  if (pst != GRX_SUCCESS) return pst;
  g->xb_blocks = pt.blocks;
  g->xb_piece = pt.piece;
  g->xb_long = pt.longrows;
  g->n_xb_pieces = pt.n_pieces;
  g->n_xb_long = pt.n_long;
  // Row blocks sorted by source (xb_sort_blocks_kernel): OPT-IN, GRX_PR_SORT_BLOCKS=1.  Measured (profiles/
  // r5_c17_pr_sorted_blocks.txt): LJ stand-in 0.486 -> 0.457 ms per iteration, kron stand-in 0.691 -> 0.690 -- for 3.1 / 7.2 ms
  // of preprocessing, i.e. more than a whole run costs: it pays only for a graph that is ranked many times.  Results are
  // bit-identical either way (same products at the same positions of the same prefix sums).
  {
    const char* sb_env = getenv("GRX_PR_SORT_BLOCKS");
    if (sb_env && *sb_env == '1' && pt.n_blocks > 0) {
      prep_timer t3("  xcd layout: row blocks sorted by source", s);
      if (!g->xb_pos) GRX_HIP(hipMalloc(reinterpret_cast<void**>(&g->xb_pos), (size_t)std::max<int64_t>(E, 1) * sizeof(uint16_t)));
      hipLaunchKernelGGL(xb_sort_blocks_kernel, dim3(std::min(pt.n_blocks, ctx->num_cus * 8)), dim3(256), 0, s,
                         reinterpret_cast<const int4*>(pt.blocks), pt.n_blocks, g->xb_ci, g->xb_w, g->xb_pos);
      GRX_HIP(hipStreamSynchronize(s));
      GRX_HIP(hipGetLastError());
    }
  }
  g->has_xb = true;
  return GRX_SUCCESS;
}

}  // namespace grx

using namespace grx;

template <class K>
static int pr_resident_per_cu(K kernel) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, PR_BLOCK, 0) != hipSuccess || n < 1) n = 4;
  (void)hipGetLastError();
  return n > 8 ? 8 : n;
}

extern "C" grx_status_t grx_pr(grx_context_t ctx, grx_graph_t g, float alpha, float tol,
                               const grx_options_t* options, float* d_p, int32_t* iterations,
                               float* elapsed_ms) {
  if (!ctx || !g || !d_p) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_pr: null argument");
  if (g->V <= 0) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_pr: empty graph");
  grx_options_t opt;
  if (options) opt = *options; else grx_options_default(&opt);
  GRX_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;

  // one-time per graph: transpose + static partition (graph preparation, like
  // the CSR build it is outside the timed enact region)
  // (threshold 12 edges per vertex since round 4: the LJ stand-in, 14.2, runs 0.50 instead of 0.78 ms per iteration blocked)
  // Layout: dense graphs whose gathered vector exceeds one XCD's L2 use the XCD-blocked
  // buckets (8 partial sums per row are cheap next to E gathers); sparse graphs (the 8 V
  // row visits would rival E) and small ones keep the plain transpose.
  // engine_flags: 0x40 = never, 0x80 = always (tests / A-B runs)
  const bool xcd_blocked = (opt.engine_flags & GRX_FLAG_PR_XCD_LAYOUT) != 0 ||
                           ((long long)g->E >= 12ll * g->V && (size_t)g->V * sizeof(float) > ((size_t)3 << 20) &&
                            !(opt.engine_flags & GRX_FLAG_PR_NO_XCD_LAYOUT));
  grx_status_t st = graph_weight_stats(ctx, g);
  if (st != GRX_SUCCESS) return st;
  // all weights exactly 1.0 (a pattern .mtx as the reference loads it): x * 1.0f == x, so the
  // weight streams are neither built nor read -- 8 E instead of 12 E bytes per iteration
  const bool unit = graph_unit_weights(g);
  if (xcd_blocked) {
    st = build_pr_xcd_layout(ctx, g);
    if (st != GRX_SUCCESS) return st;
  } else {
    st = graph_build_transpose(ctx, g);
    if (st != GRX_SUCCESS) return st;
    st = build_pr_partition(g);
    if (st != GRX_SUCCESS) return st;
  }

  const size_t V = (size_t)g->V;
  // dangling-mass partials: one per prepare block; the XCD-blocked path prepares inside pr_combine_kernel (its grid)
  const int combine_grid = ctx->num_cus * 4;
  const int n_partial = xcd_blocked ? combine_grid : std::min<int>(2048, (int)((V + PR_BLOCK - 1) / PR_BLOCK));
  GRX_HIP(ctx->fbuf[0].reserve(((V + XB - 1) / XB) * XB * sizeof(float)));  // x (permuted positions reach 8 * ceil(V / 8))
  GRX_HIP(ctx->fbuf[1].reserve(V * sizeof(float)));  // iweights
  GRX_HIP(ctx->fbuf[2].reserve(((size_t)n_partial + 16) * sizeof(float)));
  GRX_HIP(ctx->fbuf[3].reserve(((size_t)std::max(g->n_pr_pieces, g->n_xb_pieces) + 16) * sizeof(float)));
  if (xcd_blocked) GRX_HIP(ctx->far[0].reserve((size_t)XB * V * sizeof(float)));  // partial sums (scratch reuse)
  GRX_HIP(ctx->misc.reserve(64));

  pr_args a{};
  a.ro = g->ro; a.w = unit ? nullptr : g->w;
  a.t_ro = g->t_ro; a.t_ci = g->t_ci; a.t_w = unit ? nullptr : g->t_w;
  a.V = g->V; a.ctrl = ctx->d_ctrl;
  a.p = d_p;
  a.x = ctx->fbuf[0].as<float>();
  a.iw = ctx->fbuf[1].as<float>();
  a.partial = ctx->fbuf[2].as<float>();
  a.piece_sum = ctx->fbuf[3].as<float>();
  a.blocks = reinterpret_cast<const int4*>(g->pr_blocks);
  a.piece = g->pr_piece;
  a.longrows = g->pr_long;
  a.n_blocks = g->n_pr_blocks; a.n_long = g->n_pr_long; a.n_partial = n_partial;
  a.alpha = alpha; a.tol = tol;
  a.err_bits = ctx->misc.as<unsigned>();
  a.base = reinterpret_cast<float*>(ctx->misc.as<unsigned>() + 4);
  a.xb_ro = g->xb_ro; a.xb_ci = g->xb_ci; a.xb_w = unit ? nullptr : g->xb_w;
  a.xb_pos = g->xb_pos;
  a.xb = xcd_blocked ? g->xb_n : XB;
  a.xb_per_block = (g->V + a.xb - 1) / a.xb;
  a.xb_blocks = reinterpret_cast<const int4*>(g->xb_blocks);
  a.xb_piece = g->xb_piece; a.xb_long = g->xb_long;
  for (int i = 0; i <= XB; ++i) a.xb_begin[i] = g->xb_begin[i];
  a.n_xb_long = g->n_xb_long;
  a.partial_y = xcd_blocked ? ctx->far[0].as<float>() : nullptr;
  a.x_perm = xcd_blocked ? g->xb_perm : nullptr;
  a.row_lo = 0; a.row_hi = g->V; a.n_ranks = 1; a.scal_out = nullptr; a.gathered = nullptr;

  // problem.reset() (pr.hxx:65-93), outside the timed region as in the reference
  GRX_HIP(fill_f32(s, d_p, (float)(1.0 / (double)g->V), g->V));
  hipLaunchKernelGGL(pr_iweights_kernel, dim3(1024), dim3(256), 0, s, a);

  GRX_HIP(hipEventRecord(ctx->ev_begin, s));
  hipLaunchKernelGGL(pr_init_kernel, dim3(1), dim3(64), 0, s, a);

  // Persistent workgroups stride over the blocks with gridDim: the grid is exactly what is RESIDENT (round 5: it was 8 per CU
  // while the kernels' 70 VGPRs let 7 in -- the eighth of every CU ran alone behind the others, a second round for an eighth
  // of the work).  GRX_PR_WG_PER_CU: another number (tuning aid).  The block -> workgroup map does not enter any result.
  static const int per_cu_plain = pr_resident_per_cu(pr_pull_kernel), per_cu_xcd = pr_resident_per_cu(pr_pull_xcd_kernel<false>);
  int wg_env = 0;
  if (const char* e = getenv("GRX_PR_WG_PER_CU")) wg_env = atoi(e);
  const int pull_grid = std::max(1, std::min(std::max(1, g->n_pr_blocks), ctx->num_cus * (wg_env > 0 ? wg_env : per_cu_plain)));
  // hot head of every slice in LDS (pr_pull_xcd_kernel): GRX_PR_HOT entries (default 3072 = 12 KB: five workgroups per CU; measured: profiles/r5_c19_pr_hot_head_in_lds.txt, r5_c20_pr_hot_head_sizes.txt);
  // needs the hub-first order of the gathered vector
  int hot_n = 0, xcd_wg = wg_env > 0 ? wg_env : per_cu_xcd;
  if (xcd_blocked && g->xb_perm) {
    hot_n = 3072;
    if (const char* e = getenv("GRX_PR_HOT")) hot_n = atoi(e);
    hot_n = std::max(0, std::min(hot_n, std::min((g->V + a.xb - 1) / a.xb, 36864)));
    if (hot_n > 0) {
      const size_t dyn = (size_t)hot_n * sizeof(float);
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(pr_pull_xcd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
        (void)hipGetLastError();
        hot_n = 0;
      } else if (wg_env <= 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pr_pull_xcd_kernel<true>, PR_BLOCK, dyn) == hipSuccess && n >= 1) xcd_wg = std::min(n, 8);
        (void)hipGetLastError();
      }
    }
  }
  a.hot_n = hot_n;
  const int xcd_grid = std::max(a.xb, ctx->num_cus * xcd_wg / a.xb * a.xb);
  const int max_iter = opt.max_iterations > 0 ? opt.max_iterations : 0x7fffffff;
  // GRX_FLAG_PROFILE: one record per iteration from events on this stream -- advance_ms = the pull
  // (the SpMV-shaped gather incl. long-row pieces and the combine), other_ms = prepare + scalar
  const bool profile = (opt.engine_flags & GRX_FLAG_PROFILE) != 0;
  hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
  if (profile) for (auto& e : pe) GRX_HIP(hipEventCreate(&e));
  ctx->levels.clear();
  // First blind batch: as many iterations as the previous run on this graph handle needed + the one that finds the norm below
  // the tolerance (the count is a function of graph, alpha and tol: a repeated run is predicted exactly), instead of 4 + 8
  // with a host round trip in between and the rest of the second batch exiting at once.  GRX_GROUP_HINT=0: off
  const char* hint_env = getenv("GRX_GROUP_HINT");
  const bool use_hint = !(hint_env && *hint_env == '0');
  const int hinted = (use_hint && !profile) ? g->pr_iter_hint.load(std::memory_order_relaxed) : 0;
  int launched = 0, batch = profile ? 1 : (hinted > 0 ? std::min(std::max(hinted + 1, 4), 128) : 4);
  bool first_is_hint = hinted > 0;
  for (;;) {
    for (int i = 0; i < batch && launched < max_iter; ++i, ++launched) {
      if (profile) (void)hipEventRecord(pe[0], s);
      // (XCD-blocked path: x and the dangling partials of this iteration were written by the previous iteration's
      // combine kernel, or by the one prepare launch ahead of the loop)
      if (!xcd_blocked || launched == 0) hipLaunchKernelGGL(pr_prepare_kernel, dim3(n_partial), dim3(PR_BLOCK), 0, s, a);
      hipLaunchKernelGGL(pr_scalar_kernel, dim3(1), dim3(PR_BLOCK), 0, s, a, launched);
      if (profile) (void)hipEventRecord(pe[1], s);
      if (xcd_blocked) {
        if (a.hot_n > 0)
          hipLaunchKernelGGL(pr_pull_xcd_kernel<true>, dim3(xcd_grid), dim3(PR_BLOCK), (size_t)a.hot_n * sizeof(float), s, a);
        else
          hipLaunchKernelGGL(pr_pull_xcd_kernel<false>, dim3(xcd_grid), dim3(PR_BLOCK), 0, s, a);
        if (g->n_xb_long > 0)
          hipLaunchKernelGGL(pr_long_xcd_kernel, dim3((g->n_xb_long + 255) / 256), dim3(256), 0, s, a);
        hipLaunchKernelGGL(pr_combine_kernel, dim3(combine_grid), dim3(256), 0, s, a, launched);
      } else {
        hipLaunchKernelGGL(pr_pull_kernel, dim3(pull_grid), dim3(PR_BLOCK), 0, s, a, launched);
        if (g->n_pr_long > 0)
          hipLaunchKernelGGL(pr_long_kernel, dim3((g->n_pr_long + 255) / 256), dim3(256), 0, s, a, launched);
      }
      if (profile) {
        (void)hipEventRecord(pe[2], s);
        (void)hipEventSynchronize(pe[2]);
        level_rec r{};
        (void)hipEventElapsedTime(&r.other_ms, pe[0], pe[1]);
        (void)hipEventElapsedTime(&r.advance_ms, pe[1], pe[2]);
        r.frontier_size = g->V;
        r.edges = g->E;
        ctx->levels.push_back(r);
      }
    }
    GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, s));
    GRX_HIP(hipStreamSynchronize(s));
    if (profile && ctx->h_ctrl->done && !ctx->levels.empty() && (int)ctx->levels.size() > ctx->h_ctrl->pr_iter)
      ctx->levels.pop_back();  // the group that only detected convergence
    if (ctx->h_ctrl->done || launched >= max_iter) break;
    if (first_is_hint) { batch = 4; first_is_hint = false; }
    else if (!profile && batch < 16) batch *= 2;
  }
  if (ctx->h_ctrl->done && opt.max_iterations <= 0) g->pr_iter_hint.store(ctx->h_ctrl->pr_iter, std::memory_order_relaxed);
  if (profile) for (auto& e : pe) (void)hipEventDestroy(e);
  GRX_HIP(hipGetLastError());
  GRX_HIP(hipEventRecord(ctx->ev_end, s));
  GRX_HIP(hipEventSynchronize(ctx->ev_end));
  float ms = 0;
  GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  const int iters = ctx->h_ctrl->pr_iter;  // loop() executions (set by pr_scalar_kernel)
  ctx->stats.edges_visited = (int64_t)g->E * iters;
  ctx->stats.vertices_visited = (int64_t)g->V * iters;
  ctx->stats.search_depth = iters;
  ctx->stats.elapsed_ms = ms;
  ctx->stats.n_levels_recorded = (int32_t)ctx->levels.size();
  if (iterations) *iterations = iters;
  if (elapsed_ms) *elapsed_ms = ms;
  return GRX_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------
// Partitioned PageRank: rank r owns the vertices [lo, hi) -- their out-rows (for the inverse weight
// sums) and their in-rows (what the pull gathers over), both as V-row CSRs with the other rows empty and global
// column ids, i.e. what grx_host_csr_generate_rows / _in_rows produce.  One iteration on every rank:
//   pre : x[v] = p[v] * iw[v] for the owned v, written into this rank's slice of the GLOBAL x buffer; the rank's
//         {dangling sum, norm of the previous iteration} pair
//   (caller) all-gather of the x slices (S floats per rank) and of the pairs (2 words per rank)
//   post: convergence test + base term from the gathered pairs (identical on every rank), pull over the owned rows
// p is SHARDED: d_p_local holds the owned slice only (vertex v at v - lo).  The reference has no multi-GPU PageRank
// (every operator throws for context.size() != 1: advance/advance.hxx:129-132); the recurrence is pr.hxx:107-195.
struct grx_pr_dist {
  grx_context_t ctx = nullptr;
  grx_graph_t g_out = nullptr, g_in = nullptr;
  int32_t rank = 0, n_ranks = 1, lo = 0, hi = 0;
  pr_partition part;
  pr_args a{};
  int32_t launched = 0;
  bool active = false;
};

extern "C" {

grx_status_t grx_pr_dist_create(grx_context_t ctx, grx_graph_t out_rows, grx_graph_t in_rows, int32_t n_ranks,
                                int32_t my_rank, int32_t lo, int32_t hi, grx_pr_dist** out) {
  if (!ctx || !out_rows || !in_rows || !out) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_pr_dist_create: null argument");
  if (out_rows->V != in_rows->V || n_ranks < 1 || my_rank < 0 || my_rank >= n_ranks || lo < 0 || hi < lo || hi > out_rows->V)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_pr_dist_create: inconsistent partition");
  GRX_HIP(hipSetDevice(ctx->device));
  auto* h = new grx_pr_dist();
  h->ctx = ctx; h->g_out = out_rows; h->g_in = in_rows;
  h->rank = my_rank; h->n_ranks = n_ranks; h->lo = lo; h->hi = hi;
  std::vector<int32_t> ro((size_t)in_rows->V + 1);
  GRX_HIP(hipMemcpy(ro.data(), in_rows->ro, ro.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  grx_status_t st = build_pr_partition_rows(ro, lo, hi, &h->part);
  if (st != GRX_SUCCESS) { delete h; return st; }
  *out = h;
  return GRX_SUCCESS;
}

// d_p_local: hi - lo floats (the owned slice); d_x: the global x buffer (>= V floats, this rank writes [lo, hi));
// d_pair_out: 2 words; d_pairs: 2 * n_ranks words (the all-gathered pairs)
grx_status_t grx_pr_dist_begin(grx_pr_dist* h, float alpha, float tol, float* d_p_local, float* d_x,
                               uint32_t* d_pair_out, const uint32_t* d_pairs) {
  if (!h || !d_p_local || !d_x || !d_pair_out || !d_pairs) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_pr_dist_begin: null argument");
  grx_context_t ctx = h->ctx;
  GRX_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const bool unit = !h->g_in->w;  // (uniform 1.0 weights on ONE rank say nothing about the others: only "no values")
  const size_t V = (size_t)h->g_out->V;
  const int n_rows = h->hi - h->lo;
  const int n_partial = std::max(1, std::min<int>(2048, (n_rows + PR_BLOCK - 1) / PR_BLOCK));
  GRX_HIP(ctx->fbuf[1].reserve(V * sizeof(float)));  // iweights (owned rows are the ones that matter)
  GRX_HIP(ctx->fbuf[2].reserve(((size_t)n_partial + 16) * sizeof(float)));
  GRX_HIP(ctx->fbuf[3].reserve(((size_t)h->part.n_pieces + 16) * sizeof(float)));
  GRX_HIP(ctx->misc.reserve(64));
  pr_args a{};
  a.ro = h->g_out->ro; a.w = unit ? nullptr : h->g_out->w;
  a.t_ro = h->g_in->ro; a.t_ci = h->g_in->ci; a.t_w = unit ? nullptr : h->g_in->w;
  a.V = (int32_t)V; a.ctrl = ctx->d_ctrl;
  a.p = d_p_local - h->lo;  // indexed by global row: only [lo, hi) is touched
  a.x = d_x;
  a.iw = ctx->fbuf[1].as<float>();
  a.partial = ctx->fbuf[2].as<float>();
  a.piece_sum = ctx->fbuf[3].as<float>();
  a.blocks = reinterpret_cast<const int4*>(h->part.blocks);
  a.piece = h->part.piece;
  a.longrows = h->part.longrows;
  a.n_blocks = h->part.n_blocks; a.n_long = h->part.n_long; a.n_partial = n_partial;
  a.alpha = alpha; a.tol = tol;
  a.err_bits = ctx->misc.as<unsigned>();
  a.base = reinterpret_cast<float*>(ctx->misc.as<unsigned>() + 4);
  a.row_lo = h->lo; a.row_hi = h->hi; a.n_ranks = h->n_ranks;
  a.scal_out = d_pair_out; a.gathered = d_pairs;
  h->a = a;
  if (n_rows > 0) GRX_HIP(fill_f32(s, d_p_local, (float)(1.0 / (double)V), n_rows));
  hipLaunchKernelGGL(pr_iweights_kernel, dim3(1024), dim3(256), 0, s, a);
  GRX_HIP(hipEventRecord(ctx->ev_begin, s));
  hipLaunchKernelGGL(pr_init_kernel, dim3(1), dim3(64), 0, s, a);
  h->launched = 0;
  h->active = true;
  return GRX_SUCCESS;
}

grx_status_t grx_pr_dist_pre(grx_pr_dist* h) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_pr_dist_pre: no run in progress");
  hipStream_t s = h->ctx->stream;
  hipLaunchKernelGGL(pr_prepare_kernel, dim3(h->a.n_partial), dim3(PR_BLOCK), 0, s, h->a);
  hipLaunchKernelGGL(pr_dist_pack_kernel, dim3(1), dim3(PR_BLOCK), 0, s, h->a, h->launched);
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

grx_status_t grx_pr_dist_post(grx_pr_dist* h) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_pr_dist_post: no run in progress");
  hipStream_t s = h->ctx->stream;
  hipLaunchKernelGGL(pr_scalar_kernel, dim3(1), dim3(PR_BLOCK), 0, s, h->a, h->launched);
  static const int per_cu_plain = pr_resident_per_cu(pr_pull_kernel);
  const int pull_grid = std::max(1, std::min(std::max(1, h->part.n_blocks), h->ctx->num_cus * per_cu_plain));
  hipLaunchKernelGGL(pr_pull_kernel, dim3(pull_grid), dim3(PR_BLOCK), 0, s, h->a, h->launched);
  if (h->part.n_long > 0)
    hipLaunchKernelGGL(pr_long_kernel, dim3((h->part.n_long + 255) / 256), dim3(256), 0, s, h->a, h->launched);
  ++h->launched;
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

// synchronises the stream; *done != 0 once the run has converged (the same iteration on every rank)
grx_status_t grx_pr_dist_poll(grx_pr_dist* h, int32_t* done, int32_t* iterations) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_pr_dist_poll: no run in progress");
  grx_context_t ctx = h->ctx;
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, ctx->stream));
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  if (done) *done = ctx->h_ctrl->done;
  if (iterations) *iterations = ctx->h_ctrl->pr_iter;
  return GRX_SUCCESS;
}

grx_status_t grx_pr_dist_end(grx_pr_dist* h, grx_run_stats_t* stats) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_pr_dist_end: no run in progress");
  grx_context_t ctx = h->ctx;
  hipStream_t s = ctx->stream;
  GRX_HIP(hipEventRecord(ctx->ev_end, s));
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, s));
  GRX_HIP(hipEventSynchronize(ctx->ev_end));
  GRX_HIP(hipStreamSynchronize(s));
  float ms = 0;
  GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  const int iters = ctx->h_ctrl->pr_iter;
  ctx->stats = grx_run_stats_t{};
  ctx->stats.edges_visited = (int64_t)h->g_in->E * iters;  // this rank's share
  ctx->stats.vertices_visited = (int64_t)(h->hi - h->lo) * iters;
  ctx->stats.search_depth = iters;
  ctx->stats.elapsed_ms = ms;
  if (stats) *stats = ctx->stats;
  h->active = false;
  return GRX_SUCCESS;
}

grx_status_t grx_pr_dist_destroy(grx_pr_dist* h) {
  if (!h) return GRX_SUCCESS;
  (void)hipSetDevice(h->ctx->device);
  (void)hipStreamSynchronize(h->ctx->stream);
  if (h->part.blocks) (void)hipFree(h->part.blocks);
  if (h->part.piece) (void)hipFree(h->part.piece);
  if (h->part.longrows) (void)hipFree(h->part.longrows);
  delete h;
  return GRX_SUCCESS;
}

}  // extern "C"
