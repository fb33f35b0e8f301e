// grx_block.hip -- BLOCK-ASYNCHRONOUS relaxation for road-like graphs (BFS and non-negative SSSP), round 4.
//
// What it replaces in the reference: the same enactor loop as everything else here -- one advance (+ filter) per hop
// (algorithms/bfs.hxx:87-147, sssp.hxx:104-159, framework/enactor.hxx:243-288) -- for graphs where that loop is a chain of
// thousands of dependent, nearly empty levels (road_usa / its stand-in: ~5000 levels of ~11 k edges; the level-synchronous
// floor of this engine is 7.2 us per level = 36 ms, 0.35 % of the HBM roofline).
//
// Idea.  The fixed point of label-correcting relaxation does not depend on the schedule, so relax
// ASYNCHRONOUSLY inside spatially local blocks and synchronise only between them:
//   * once per graph (cached in the handle): the vertices are cut into BLOCKS of <= NV vertices by breadth-first region
//     growing (host, multi-threaded) and renumbered block by block.  A block's INTRA-block edges are stored as 16-bit local
//     targets behind 16-bit offsets -- small enough that the whole block (labels, offsets, edges[, weights]) lives in LDS;
//     the few edges that leave the block (~1.5 % on the road stand-in) form an ordinary CSR over the new ids.
//   * a search keeps two label arrays in the new numbering: dist (the tentative label, lowered by atomics from anywhere) and
//     expd (the label a vertex was last EXPANDED with).  dist[v] < expd[v] means v is pending.  bmin[b] is a lower bound of
//     the pending labels of block b.
//   * a SUPERSTEP is two launches: the head kernel (one workgroup) selects the ACTIVE blocks -- bmin[b] below the bound
//     `hi` of the current global bucket [lo, hi) -- or, when there is none, opens the next bucket at the smallest pending
//     label (delta-stepping over blocks: labels below `hi` are final once the bucket has drained, so wrong labels cannot
//     flood the graph); the block kernel runs one workgroup per active block: it stages the block in LDS, relaxes it to its
//     LOCAL fixed point -- dozens to hundreds of hops, each a handful of LDS operations and ONE barrier, no global memory --
//     expanding only labels < hi, then relaxes the boundary edges of the vertices it expanded (atomicMin on dist of the
//     neighbour block + atomicMin on that block's bmin), writes its labels back and leaves the minimum of what is still
//     pending in bmin[b].
//   A CPU prototype of this schedule (tools/proto/block_async.c) on a 4 M-vertex stand-in: 127 supersteps instead of 2050
//   levels at 2.2x the relaxations (bucket of 256 hops, blocks of 8192); weighted U{1..1000}: 248 supersteps at 4.2x.
//
// Correctness.  Every relaxation is the reference's (BFS: depth + 1 with atomicMin, bfs.hxx:105-119; SSSP: fl(d + w) with
// atomicMin, sssp.hxx:116-126); labels are compared as 32-bit keys (depths, or the bit patterns of non-negative floats: the
// same order).  The search ends when no block has a pending label: then dist[v] <= fl(dist[u] + w) for every edge and every
// finite label is the label of a path from the source -- the fixed point the reference's loop and its CPU Dijkstra reach
// (bit for bit: fl(a + w) is monotone in a).
// Memory model: a block is touched by ONE workgroup per launch; what other workgroups do to it in the same launch are
// device-scope atomics (performed at the memory side), its own write-backs are device-scope atomics / write-through stores,
// and everything a launch reads that an earlier launch wrote crosses a kernel boundary.  Notifications cannot be lost: the
// owner swaps bmin[b] to "none" and waits for the answer BEFORE it loads the labels; a neighbour lowers dist first and bmin
// after that atomic has returned.
#include "grx_engine.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <vector>

namespace grx {

constexpr uint32_t BLK_NONE = 0xffffffffu;
#ifndef GRX_BLOCK_DEFAULT
#define GRX_BLOCK_DEFAULT 0  // round 4, call 2: correct everywhere, but 120-160 ms against 40 ms level-synchronous on the road stand-in
#endif

// per graph (device arrays; built once, cached in the graph handle)
struct blk_graph {
  int32_t nb = 0, nv = 0, ne = 0, offs = 0;  // blocks; vertices / intra-edge slots per block; stride of the offset array
  int32_t n_new = 0;                          // nb * nv ids in the new numbering
  int32_t weighted = 0;
  int64_t n_cross = 0;
  int32_t* perm = nullptr;        // original id -> new id [V]
  int32_t* inv = nullptr;         // new id -> original id [n_new], -1 = padding
  unsigned short* off = nullptr;  // [nb][offs]: offsets of the intra-block edges of the block's vertices
  unsigned short* tgt = nullptr;  // [nb][ne]: local target
  float* wt = nullptr;            // [nb][ne] (weighted)
  int32_t* nedge = nullptr;       // [nb] intra-block edges
  int32_t* xro = nullptr;         // [n_new + 1] offsets of the block-leaving edges, new numbering
  int32_t* xci = nullptr;         // targets (new ids)
  float* xw = nullptr;
  uint32_t* bnd = nullptr;        // [nb][nv / 32]: the vertex has an in-edge from another block
  uint32_t* xout = nullptr;       // [nb][nv / 32]: the vertex has an out-edge that leaves its block
  double host_ms = 0.0;           // what the build cost (reported, not timed)
};

struct blk_dev {
  int32_t nb, nv_shift, n_new;
  const int32_t* inv;
  const unsigned short* off;
  const unsigned short* tgt;
  const float* wt;
  const int32_t* nedge;
  const int32_t* xro;
  const int32_t* xci;
  const float* xw;
  const uint32_t* bnd;
  const uint32_t* xout;
};

struct blk_run {
  uint32_t* dist;
  uint32_t* expd;
  uint32_t* bmin;
  int32_t* queue;
  ctrl_t* ctrl;
  int32_t* mailbox;
  uint32_t inf;    // key of "unreached": INT_MAX (BFS) / bits of FLT_MAX (SSSP)
  uint32_t delta;  // bucket width: hops (BFS) / bits of a float (SSSP)
};

__device__ __forceinline__ uint32_t blk_key_add(uint32_t key, uint32_t delta, bool weighted, uint32_t inf) {
  if (!weighted) {
    const unsigned long long s = (unsigned long long)key + delta;
    return s < inf ? (uint32_t)s : inf;
  }
  const float a = __uint_as_float(key);
  float f = a + __uint_as_float(delta);
  if (!(f > a)) f = nextafterf(a, FLT_MAX);  // a bucket always moves on
  const uint32_t k = __float_as_uint(f);
  return k < inf ? k : inf;
}

__global__ void blk_reset_kernel(blk_run r, int64_t n_new, int32_t nb) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint4 v4 = make_uint4(r.inf, r.inf, r.inf, r.inf);
  uint4* d4 = reinterpret_cast<uint4*>(r.dist);
  uint4* e4 = reinterpret_cast<uint4*>(r.expd);
  for (int64_t i = i0; i < n_new / 4; i += stride) {  // n_new is a multiple of nv >= 1024
    d4[i] = v4;
    e4[i] = v4;
  }
  for (int64_t i = i0; i < nb; i += stride) r.bmin[i] = BLK_NONE;
}

__global__ void blk_seed_kernel(blk_run r, blk_dev g, const int32_t* perm, int32_t src, int weighted) {
  if (threadIdx.x != 0) return;
  const int32_t s = perm[src];
  r.dist[s] = 0u;
  r.bmin[s >> g.nv_shift] = 0u;
  ctrl_t* c = r.ctrl;
  c->level = 0;
  c->done = 0;
  c->edges_visited = 0;
  c->vertices_visited = 0;
  c->mode = 0;
  c->mid_err = 0;
  c->spare[0] = (int32_t)blk_key_add(0u, r.delta, weighted != 0, r.inf);  // first bucket [0, delta)
  c->spare[1] = 0;
  c->spare[2] = 1;
  c->spare[3] = 0;
  c->bu_open = 0;
  c->bu_probes = 0;
  c->t_start = (long long)wall_clock64();
  r.mailbox[0] = 0;
}

// Head of a superstep, ONE workgroup: which blocks are active in the current bucket; when none is, the next bucket opens at
// the smallest pending label; when nothing is pending the search is over.  <<<1, 1024>>>
__global__ __launch_bounds__(1024) void blk_head_kernel(blk_run r, blk_dev g, int weighted) {
  __shared__ int s_wave[1024 / 64 + 1];
  __shared__ unsigned s_min[16];
  __shared__ int s_cnt[16];
  __shared__ unsigned s_hi;
  ctrl_t* c = r.ctrl;
  if (c->done) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  unsigned hi = (unsigned)c->spare[0];
  unsigned mn = BLK_NONE;
  int cnt = 0;
  for (int b = tid; b < g.nb; b += 1024) {
    const unsigned m = r.bmin[b];
    mn = min(mn, m);
    cnt += m < hi ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = min(mn, (unsigned)__shfl_xor((int)mn, o, 64));
    cnt += __shfl_xor(cnt, o, 64);
  }
  if (lane == 0) { s_min[wid] = mn; s_cnt[wid] = cnt; }
  __syncthreads();
  if (tid == 0) {
    unsigned m = BLK_NONE;
    int n = 0;
    for (int i = 0; i < 16; ++i) { m = min(m, s_min[i]); n += s_cnt[i]; }
    if (n == 0 && m != BLK_NONE) {  // the bucket has drained: the next one starts at the smallest pending label
      hi = blk_key_add(m, r.delta, weighted != 0, r.inf);
      if (hi <= m) hi = m + 1u;  // (labels at the very top of the key range)
      c->spare[0] = (int32_t)hi;
      c->spare[2] += 1;
    }
    s_hi = (n == 0 && m == BLK_NONE) ? 0u : hi;
    if (n == 0 && m == BLK_NONE) {
      c->done = 1;
      long long* mb64 = reinterpret_cast<long long*>(r.mailbox + 4);
      mb64[0] = c->edges_visited;
      mb64[1] = c->vertices_visited;
      mb64[2] = (long long)wall_clock64() - c->t_start;
      r.mailbox[1] = c->level;
      __threadfence_system();
      r.mailbox[0] = 1;
    }
  }
  __syncthreads();
  hi = s_hi;
  if (hi == 0u) return;  // done
  int base = 0;
  for (int b0 = 0; b0 < g.nb; b0 += 1024) {
    const int b = b0 + tid;
    const int act = (b < g.nb && r.bmin[b] < hi) ? 1 : 0;
    int tot;
    const int ex = dev::block_exclusive_sum<1024>(act, s_wave, &tot);
    if (act) r.queue[base + ex] = b;
    base += tot;
  }
  if (tid == 0) {
    c->spare[1] = base;
    c->level += 1;
    c->vertices_visited += base;  // block activations
  }
}

template <int NV, int NE, bool W>
struct blk_smem {
  static constexpr int WORDS = NV / 32;
  static constexpr int OFFS = NV + 8;
  alignas(16) uint32_t lab[NV];
  alignas(16) unsigned short off[OFFS];
  alignas(16) unsigned short tgt[NE];
  alignas(16) float wt[W ? NE : 4];
  uint32_t fb[2][WORDS];
  uint32_t chg[WORDS];
  uint32_t pend[WORDS];
  uint32_t bnd[WORDS];
  uint32_t xout[WORDS];
  int flag[3];
  unsigned left;
  int nrelax;
  unsigned exch;
};

// One superstep: workgroup i takes the active blocks i, i + gridDim.x, ...  NV / 32 threads: thread t owns word t of the
// block's bitmaps (32 vertices).
template <int NV, int NE, bool W>
__global__ __launch_bounds__(NV / 32) void blk_kernel(blk_run r, blk_dev g) {
  using S = blk_smem<NV, NE, W>;
  constexpr int T = NV / 32;
  static_assert(T % 64 == 0 && T >= 64, "whole waves; one bitmap word per thread");
  __shared__ S sm;
  ctrl_t* c = r.ctrl;
  if (__hip_atomic_load(&c->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const int n_active = c->spare[1];
  const uint32_t hi = (uint32_t)c->spare[0];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int item = (int)blockIdx.x; item < n_active; item += (int)gridDim.x) {
    const int b = r.queue[item];
    const size_t base = (size_t)b * NV;
    // ---- the notification word first (its answer must be in before the labels are read) ...
    if (tid == 0) {
      sm.exch = __hip_atomic_exchange(&r.bmin[b], BLK_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sm.flag[0] = 0; sm.flag[1] = 0; sm.flag[2] = 0;
      sm.left = BLK_NONE;
      sm.nrelax = 0;
    }
    // ---- ... then the immutable part of the block: offsets, local targets, weights, the two boundary bitmaps.  Whole
    // capacity (the unused tail is zeros), so that no load waits for the block's edge count, and all of a thread's loads
    // in flight together (a copy loop waits for every load before it issues the next: 12+ serialized round trips in the
    // first version's ISA)
    {
      constexpr int N_OFF = (S::OFFS / 8 + T - 1) / T, N_TGT = NE / 8 / T, N_WT = W ? NE / 4 / T : 0;
      static_assert(NE % (8 * T) == 0, "whole 16-byte pieces per thread");
      const uint4* s4 = reinterpret_cast<const uint4*>(g.off + (size_t)b * S::OFFS);
      const uint4* t4 = reinterpret_cast<const uint4*>(g.tgt + (size_t)b * NE);
      uint4 vo[N_OFF], vt[N_TGT];
#pragma unroll
      for (int j = 0; j < N_OFF; ++j) vo[j] = s4[min(j * T + tid, S::OFFS / 8 - 1)];
#pragma unroll
      for (int j = 0; j < N_TGT; ++j) vt[j] = t4[j * T + tid];
      const uint32_t vb = g.bnd[(size_t)b * S::WORDS + tid], vx = g.xout[(size_t)b * S::WORDS + tid];
      uint4* d4 = reinterpret_cast<uint4*>(sm.off);
      uint4* td4 = reinterpret_cast<uint4*>(sm.tgt);
#pragma unroll
      for (int j = 0; j < N_OFF; ++j)
        if (j * T + tid < S::OFFS / 8) d4[j * T + tid] = vo[j];
#pragma unroll
      for (int j = 0; j < N_TGT; ++j) td4[j * T + tid] = vt[j];
      if constexpr (W) {
        const uint4* w4 = reinterpret_cast<const uint4*>(g.wt + (size_t)b * NE);
        uint4* wd4 = reinterpret_cast<uint4*>(sm.wt);
#pragma unroll
        for (int h0 = 0; h0 < N_WT; h0 += 12) {
          uint4 vw[12];
#pragma unroll
          for (int j = 0; j < 12; ++j) vw[j] = w4[min(h0 + j, N_WT - 1) * T + tid];
#pragma unroll
          for (int j = 0; j < 12; ++j)
            if (h0 + j < N_WT) wd4[(h0 + j) * T + tid] = vw[j];
        }
      }
      sm.bnd[tid] = vb;
      sm.xout[tid] = vx;
      sm.fb[1][tid] = 0u;
    }
    __syncthreads();  // thread 0 is past its exchange (it stored the answer)
    // ---- labels: lab <- dist, pending = dist < expd, seeds = pending below the bucket bound
    {
      const uint32_t* dg = r.dist + base;
      const uint32_t* eg = r.expd + base;
      // (eight labels and eight expansion marks per thread in flight: written iteration by iteration the loop waited for
      // every pair of loads before it issued the next -- 32 serialized round trips per activation in the first version's ISA)
#pragma unroll 1
      for (int k0 = 0; k0 < 32; k0 += 8) {
        uint32_t dv[8], ev[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          dv[j] = dg[(k0 + j) * T + tid];
          ev[j] = eg[(k0 + j) * T + tid];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int v = (k0 + j) * T + tid;
          const uint32_t d = dv[j], e = ev[j];
          sm.lab[v] = d;
          const bool pnd = d < e;
          const bool sd = pnd && d < hi;
          const unsigned long long bp = dev::ballot(pnd), bs = dev::ballot(sd);
          const int w0 = v >> 5;  // lanes 0..31: word w0, lanes 32..63: the next one
          if (lane == 0) {
            sm.pend[w0] = (uint32_t)bp;
            sm.fb[0][w0] = (uint32_t)bs;
            sm.chg[w0] = (uint32_t)bs;
            if (bs) sm.flag[0] = 1;
          } else if (lane == 32) {
            sm.pend[w0] = (uint32_t)(bp >> 32);
            sm.fb[0][w0] = (uint32_t)(bs >> 32);
            sm.chg[w0] = (uint32_t)(bs >> 32);
          }
        }
      }
    }
    // ---- local fixed point: a round expands the vertices of the current frontier bitmap; one barrier per round
    int nrel = 0;
    for (int rd = 0;; ++rd) {
      __syncthreads();
      const int go = sm.flag[rd % 3];
      uint32_t wv = sm.fb[rd & 1][tid];  // (issued together with the flag: one round trip)
      if (!go) break;
      if (tid == 0) sm.flag[(rd + 2) % 3] = 0;  // read last two rounds ago, written next round
      sm.fb[rd & 1][tid] = 0u;  // written again from the next round on (behind the next barrier)
      uint32_t* nxt = sm.fb[(rd + 1) & 1];
      while (wv) {
        const int v = tid * 32 + __builtin_ctz(wv);
        wv &= wv - 1u;
        // A round is a chain of DEPENDENT LDS round trips (~100+ cycles each with two to four waves on the CU): label +
        // offsets -> targets -> atomic min -> (fire-and-forget marks).  The first version walked the edges one by one
        // (target, min, marks: three trips per edge, 2.2 us per round on the road stand-in); here the operations of up
        // to four edges are issued together, from clamped indices, so a vertex costs one chain whatever its degree.
        uint32_t lu = sm.lab[v];
        const int e0 = sm.off[v], e1 = sm.off[v + 1];
        asm volatile("" : "+v"(lu));  // (the label travels with the offsets, not behind the branch on them)
        nrel += e1 - e0;
        for (int eb = e0; eb < e1; eb += 4) {
          int t[4];
          uint32_t cand[4], old[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = min(eb + j, e1 - 1);
            t[j] = sm.tgt[e];
            if constexpr (W) cand[j] = __float_as_uint(__uint_as_float(lu) + sm.wt[e]);
            else cand[j] = lu + 1u;
            if (eb + j >= e1) cand[j] = BLK_NONE;  // no edge: a min with the largest key changes nothing
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) old[j] = atomicMin(&sm.lab[t[j]], cand[j]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (cand[j] < old[j]) {
              const uint32_t bit = 1u << (t[j] & 31);
              atomicOr(&sm.chg[t[j] >> 5], bit);
              if (cand[j] < hi) {
                atomicOr(&nxt[t[j] >> 5], bit);
                sm.flag[(rd + 1) % 3] = 1;
              }
            }
          }
        }
      }
    }
    // ---- out.  Labels back and expansion marks, COALESCED (lane <-> consecutive vertex, like the staging; the first two
    // versions stored from the owner thread of each bitmap word -- scattered 4-byte write-through stores, the slowest
    // kind per byte): expd is the owner's alone and is rewritten whole -- L where the vertex is expanded (or was never
    // pending), "unreached" where it stays pending (any value above its label says so); dist only where it changed,
    // through an atomic min on boundary vertices (a neighbour may have lowered them since they were loaded).
    {
      uint32_t* dg = r.dist + base;
      uint32_t* eg = r.expd + base;
#pragma unroll 8
      for (int k = 0; k < 32; ++k) {
        const int v = k * T + tid;
        const uint32_t L = sm.lab[v];
        const uint32_t bit = 1u << (v & 31);
        const uint32_t cw = sm.chg[v >> 5], pw = sm.pend[v >> 5], bw = sm.bnd[v >> 5];
        eg[v] = (L < hi || !((cw | pw) & bit)) ? L : r.inf;
        if (cw & bit) {
          if (bw & bit) (void)__hip_atomic_fetch_min(&dg[v], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else dg[v] = L;
        }
      }
    }
    // what is left pending, and the edges that leave the block -- only from the vertices that have any (bitmap)
    {
      const uint32_t cw = sm.chg[tid], pw = sm.pend[tid];
      uint32_t m = cw | pw, xm = 0u;
      unsigned left = BLK_NONE;
      while (m) {
        const int j = __builtin_ctz(m);
        const uint32_t bit = 1u << j;
        m &= m - 1u;
        const uint32_t L = sm.lab[tid * 32 + j];
        if ((cw & bit) && L < hi) xm |= bit;
        if (L >= hi) left = min(left, L);  // pending: it was when it came in, or its label fell in here
      }
      xm &= sm.xout[tid];
      while (xm) {
        const int v = tid * 32 + __builtin_ctz(xm);
        xm &= xm - 1u;
        const uint32_t L = sm.lab[v];
        const size_t vg = base + (size_t)v;
        const int x0 = g.xro[vg], x1 = g.xro[vg + 1];
        nrel += x1 - x0;
        for (int eb = x0; eb < x1; eb += 4) {
          int t[4];
          uint32_t cand[4], old[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = min(eb + j, x1 - 1);
            t[j] = g.xci[e];
            if constexpr (W) cand[j] = __float_as_uint(__uint_as_float(L) + g.xw[e]);
            else cand[j] = L + 1u;
            if (eb + j >= x1) cand[j] = BLK_NONE;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            old[j] = __hip_atomic_fetch_min(&r.dist[t[j]], cand[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (issued behind the answers of the atomics above: a label is lowered BEFORE its block is notified)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (cand[j] < old[j])
              (void)__hip_atomic_fetch_min(&r.bmin[t[j] >> g.nv_shift], cand[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        left = min(left, (unsigned)__shfl_xor((int)left, o, 64));
        nrel += __shfl_xor(nrel, o, 64);
      }
      if (lane == 0) {
        if (left != BLK_NONE) atomicMin(&sm.left, left);
        if (nrel) atomicAdd(&sm.nrelax, nrel);
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (sm.left != BLK_NONE) (void)__hip_atomic_fetch_min(&r.bmin[b], sm.left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sm.nrelax)
        (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(&c->edges_visited), (unsigned long long)sm.nrelax,
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}

// labels back to the caller's numbering; reached vertices, their out-degree sum, the largest finite key
__global__ __launch_bounds__(256) void blk_final_kernel(blk_run r, blk_dev g, const int32_t* ro, uint32_t* out) {
  long long reached = 0, deg = 0;
  unsigned mx = 0u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < g.n_new; i += (int64_t)gridDim.x * 256) {
    const int o = g.inv[i];
    if (o < 0) continue;
    const uint32_t k = r.dist[i];
    out[o] = k;
    if (k != r.inf) {
      ++reached;
      deg += ro[o + 1] - ro[o];
      mx = max(mx, k);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    reached += __shfl_xor(reached, o, 64);
    deg += __shfl_xor(deg, o, 64);
    mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    ctrl_t* c = r.ctrl;
    if (reached) atomicAdd(reinterpret_cast<unsigned long long*>(&c->bu_open), (unsigned long long)reached);
    if (deg) atomicAdd(reinterpret_cast<unsigned long long*>(&c->bu_probes), (unsigned long long)deg);
    if (mx) atomicMax(reinterpret_cast<unsigned*>(&c->spare[3]), mx);
  }
}

}  // namespace grx

using namespace grx;

static int blk_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

void grx::blk_graph_free(void* p) {
  blk_graph* b = reinterpret_cast<blk_graph*>(p);
  if (!b) return;
  void* ptrs[] = {b->perm, b->inv, b->off, b->tgt, b->wt, b->nedge, b->xro, b->xci, b->xw, b->bnd, b->xout};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  delete b;
}

namespace {

template <typename F>
void parallel_chunks(int n_chunks, F&& f) {
  const int hw = std::max(1u, std::thread::hardware_concurrency());
  const int nt = std::min(n_chunks, std::min(hw, 64));
  if (nt <= 1) {
    for (int i = 0; i < n_chunks; ++i) f(i);
    return;
  }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t)
    pool.emplace_back([&] {
      for (int i; (i = next.fetch_add(1)) < n_chunks;) f(i);
    });
  for (auto& th : pool) th.join();
}

template <typename T>
hipError_t upload(T** dst, const std::vector<T>& src, hipStream_t s) {
  const size_t bytes = std::max<size_t>(src.size(), 4) * sizeof(T);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), bytes);
  if (e != hipSuccess) return e;
  if (!src.empty()) e = hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s);
  return e;
}

}  // namespace

// The block structure on the host: what blk_build uploads, and what the host emulation of the schedule
// (grx_debug_block_search_host, CPU test-suite) walks.
struct blk_host {
  int32_t nb = 0, nv = 0, ne = 0, n_new = 0;
  int64_t n_cross = 0;
  std::vector<int32_t> perm, inv, nedge, xro, xci;
  std::vector<unsigned short> off, tgt;
  std::vector<float> wt, xw;
  std::vector<uint32_t> bnd, xout;
};

// Cut the graph (host CSR) into blocks.  false: not applicable (a row that does not fit a block, negative weights, too much
// padding).
static bool blk_partition_host(int32_t V, int64_t E, const int32_t* ro, const int32_t* ci, const float* w, int NV, int NE,
                               blk_host& h) {
  const bool weighted = w != nullptr;
  (void)E;
  // ---- 1. region growing, independently on ranges of the vertex ids (a block never spans two ranges)
  const int n_ranges = (int)std::max<int64_t>(1, std::min<int64_t>(64, (int64_t)V / ((int64_t)NV * 8)));
  struct range_out {
    std::vector<int32_t> verts;   // vertices in block order
    std::vector<int32_t> sizes;   // vertices per block
    bool bad = false;
  };
  std::vector<range_out> ranges((size_t)n_ranges);
  parallel_chunks(n_ranges, [&](int rg) {
    const int32_t lo = (int32_t)((int64_t)V * rg / n_ranges), hi = (int32_t)((int64_t)V * (rg + 1) / n_ranges);
    range_out& o = ranges[(size_t)rg];
    o.verts.reserve((size_t)(hi - lo));
    std::vector<unsigned char> taken((size_t)(hi - lo), 0);
    size_t block_begin = 0, head = 0;  // current block = verts[block_begin ..), BFS head
    int64_t edges = 0;
    auto close_block = [&] {
      if (o.verts.size() > block_begin) o.sizes.push_back((int32_t)(o.verts.size() - block_begin));
      block_begin = o.verts.size();
      head = block_begin;
      edges = 0;
    };
    for (int32_t sd = lo; sd < hi; ++sd) {
      if (taken[(size_t)(sd - lo)]) continue;
      const int32_t dsd = ro[(size_t)sd + 1] - ro[(size_t)sd];
      if (dsd > NE) { o.bad = true; return; }
      // a block is filled with whole breadth-first regions one after the other: small components are packed together
      if ((int)(o.verts.size() - block_begin) >= NV || edges + dsd > NE) close_block();
      taken[(size_t)(sd - lo)] = 1;
      o.verts.push_back(sd);
      edges += dsd;
      while (head < o.verts.size()) {
        const int32_t u = o.verts[head++];
        for (int32_t e = ro[(size_t)u]; e < ro[(size_t)u + 1]; ++e) {
          const int32_t v = ci[(size_t)e];
          if (v < lo || v >= hi || taken[(size_t)(v - lo)]) continue;
          const int32_t dv = ro[(size_t)v + 1] - ro[(size_t)v];
          if ((int)(o.verts.size() - block_begin) >= NV || edges + dv > NE) continue;
          taken[(size_t)(v - lo)] = 1;
          o.verts.push_back(v);
          edges += dv;
        }
      }
    }
    close_block();
  });
  for (auto& o : ranges)
    if (o.bad) return false;  // a row longer than a block
  std::vector<int32_t> range_b0((size_t)n_ranges + 1, 0);
  for (int i = 0; i < n_ranges; ++i) range_b0[(size_t)i + 1] = range_b0[(size_t)i] + (int32_t)ranges[(size_t)i].sizes.size();
  const int32_t nb = range_b0[(size_t)n_ranges];
  if ((int64_t)nb * NV > (int64_t)INT32_MAX - NV || (int64_t)nb * NV > 3 * (int64_t)V + 64 * (int64_t)NV) return false;
  const int32_t n_new = nb * NV;
  const int OFFS = NV + 8, WORDS = NV / 32;
  h.nb = nb; h.nv = NV; h.ne = NE; h.n_new = n_new;
  // ---- 2. numbering
  h.perm.assign((size_t)V, -1);
  h.inv.assign((size_t)n_new, -1);
  parallel_chunks(n_ranges, [&](int rg) {
    const range_out& o = ranges[(size_t)rg];
    size_t at = 0;
    for (size_t k = 0; k < o.sizes.size(); ++k) {
      const int32_t b = range_b0[(size_t)rg] + (int32_t)k;
      for (int32_t i = 0; i < o.sizes[k]; ++i, ++at) {
        const int32_t v = o.verts[at];
        // INTERLEAVED local numbering: the i-th vertex of the block (breadth-first order: neighbours in the graph are
        // neighbours in i) becomes bit i / WORDS of bitmap word i % WORDS.  A thread of the block kernel owns one word; a
        // wave front passing through the block is a run of consecutive i, i.e. one vertex per THREAD.  (With local id = i
        // the front sat in a handful of words and their threads walked 10-30 vertices each while the others idled:
        // 2.2-3 us per round on the road stand-in, round 4 calls 2-3.)
        const int32_t li = (i % WORDS) * 32 + i / WORDS;
        h.perm[(size_t)v] = b * NV + li;
        h.inv[(size_t)b * NV + li] = v;
      }
    }
  });
  // ---- 3. per block: intra-block edges (16-bit), counts of the edges that leave, boundary marks of their targets
  h.off.assign((size_t)nb * OFFS, 0);
  h.tgt.assign((size_t)nb * NE, 0);
  if (weighted) h.wt.assign((size_t)nb * NE, 0.0f);
  h.nedge.assign((size_t)nb, 0);
  h.xro.assign((size_t)n_new + 1, 0);
  std::vector<std::atomic<unsigned char>> is_bnd((size_t)n_new);
  for (auto& x : is_bnd) x.store(0, std::memory_order_relaxed);
  std::atomic<int> negative{0};
  parallel_chunks(nb, [&](int b) {
    unsigned short* bo = &h.off[(size_t)b * OFFS];
    unsigned short* bt = &h.tgt[(size_t)b * NE];
    float* bw = weighted ? &h.wt[(size_t)b * NE] : nullptr;
    int32_t n = 0;
    for (int i = 0; i < NV; ++i) {
      bo[i] = (unsigned short)n;
      const int32_t v = h.inv[(size_t)b * NV + i];
      if (v < 0) continue;
      int32_t cross = 0;
      for (int32_t e = ro[(size_t)v]; e < ro[(size_t)v + 1]; ++e) {
        const int32_t t = h.perm[(size_t)ci[(size_t)e]];
        if (weighted && !(w[(size_t)e] >= 0.0f)) negative.store(1);
        if (t / NV == b) {
          bt[n] = (unsigned short)(t % NV);
          if (bw) bw[n] = w[(size_t)e];
          ++n;
        } else {
          ++cross;
          is_bnd[(size_t)t].store(1, std::memory_order_relaxed);
        }
      }
      h.xro[(size_t)b * NV + i + 1] = cross;  // shifted by one: becomes the exclusive prefix below
    }
    for (int i = NV; i < OFFS; ++i) bo[i] = (unsigned short)n;
    h.nedge[(size_t)b] = n;
  });
  if (negative.load()) return false;
  for (size_t i = 1; i <= (size_t)n_new; ++i) h.xro[i] += h.xro[i - 1];
  h.n_cross = h.xro[(size_t)n_new];
  h.xci.assign((size_t)h.n_cross, 0);
  if (weighted) h.xw.assign((size_t)h.n_cross, 0.0f);
  h.bnd.assign((size_t)nb * WORDS, 0u);
  h.xout.assign((size_t)nb * WORDS, 0u);
  parallel_chunks(nb, [&](int b) {
    for (int i = 0; i < NV; ++i) {
      const size_t id = (size_t)b * NV + i;
      if (is_bnd[id].load(std::memory_order_relaxed)) h.bnd[(size_t)b * WORDS + (i >> 5)] |= 1u << (i & 31);
      const int32_t v = h.inv[id];
      if (v < 0) continue;
      int32_t at = h.xro[id];
      if (h.xro[id + 1] > at) h.xout[(size_t)b * WORDS + (i >> 5)] |= 1u << (i & 31);
      for (int32_t e = ro[(size_t)v]; e < ro[(size_t)v + 1]; ++e) {
        const int32_t t = h.perm[(size_t)ci[(size_t)e]];
        if (t / NV == b) continue;
        h.xci[(size_t)at] = t;
        if (weighted) h.xw[(size_t)at] = w[(size_t)e];
        ++at;
      }
    }
  });
  return true;
}

// Build (once per graph handle and kind) the block structure.  *usable = false: the graph is not road-like enough (a vertex
// whose out-edges do not fit a block, negative weights, ...): the caller keeps its other path.
static grx_status_t blk_build(grx_context_t ctx, grx_graph_t g, bool weighted, int NV, int NE, blk_graph** out, bool* usable) {
  *usable = false;
  *out = nullptr;
  const int32_t V = g->V;
  const int64_t E = g->E;
  const auto t0 = std::chrono::steady_clock::now();
  hipStream_t s = ctx->stream;
  std::vector<int32_t> ro((size_t)V + 1), ci((size_t)E);
  std::vector<float> w;
  GRX_HIP(hipMemcpyAsync(ro.data(), g->ro, ro.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  GRX_HIP(hipMemcpyAsync(ci.data(), g->ci, ci.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (weighted) {
    w.resize((size_t)E);
    GRX_HIP(hipMemcpyAsync(w.data(), g->w, w.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  }
  GRX_HIP(hipStreamSynchronize(s));
  blk_host h;
  if (!blk_partition_host(V, E, ro.data(), ci.data(), weighted ? w.data() : nullptr, NV, NE, h)) return GRX_SUCCESS;
  blk_graph* bg = new blk_graph();
  bg->nb = h.nb; bg->nv = NV; bg->ne = NE; bg->offs = NV + 8; bg->n_new = h.n_new; bg->weighted = weighted ? 1 : 0;
  bg->n_cross = h.n_cross;
  hipError_t e = hipSuccess;
  auto up = [&](auto** dst, const auto& src) { if (e == hipSuccess) e = upload(dst, src, s); };
  up(&bg->perm, h.perm); up(&bg->inv, h.inv); up(&bg->off, h.off); up(&bg->tgt, h.tgt); up(&bg->nedge, h.nedge);
  up(&bg->xro, h.xro); up(&bg->xci, h.xci); up(&bg->bnd, h.bnd); up(&bg->xout, h.xout);
  if (weighted) { up(&bg->wt, h.wt); up(&bg->xw, h.xw); }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    blk_graph_free(bg);
    (void)hipGetLastError();
    return GRX_SUCCESS;  // not enough memory for the second copy of the graph: the caller's other path runs
  }
  bg->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *out = bg;
  *usable = true;
  return GRX_SUCCESS;
}

// HOST EMULATION of the device schedule on the structure blk_partition_host builds -- the same supersteps, buckets, local
// rounds and boundary pass as blk_head_kernel / blk_kernel, one block after the other.  Test infrastructure for the CPU
// suite (tests/test_block_host.py): it checks the partitioner and the schedule's fixed point against the oracle without a
// GPU.  No product path calls it.
extern "C" grx_status_t grx_debug_block_search_host(grx_host_csr_t csr, int32_t weighted, int32_t nv, int32_t src, uint32_t delta_bits,
                                                    uint32_t* out_keys, grx_block_stats_t* stats) {
  if (!csr || !out_keys || src < 0 || src >= csr->V) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_debug_block_search_host: bad argument");
  if (nv != 2048 && nv != 4096 && nv != 8192) return fail(GRX_ERROR_INVALID_ARGUMENT, "nv: 2048, 4096 or 8192");
  blk_host h;
  const int NV = nv, NE = 3 * nv, OFFS = NV + 8, WORDS = NV / 32;
  if (!blk_partition_host(csr->V, csr->E, csr->ro.data(), csr->ci.data(), weighted ? csr->w.data() : nullptr, NV, NE, h))
    return fail(GRX_ERROR_UNSUPPORTED, "graph does not fit the block structure");
  const uint32_t inf = weighted ? 0x7f7fffffu : (uint32_t)INT_MAX;
  auto key_add = [&](uint32_t key, uint32_t delta) -> uint32_t {
    if (!weighted) {
      const unsigned long long sum = (unsigned long long)key + delta;
      return sum < inf ? (uint32_t)sum : inf;
    }
    float a, d;
    memcpy(&a, &key, 4);
    memcpy(&d, &delta, 4);
    float f = a + d;
    if (!(f > a)) f = std::nextafterf(a, FLT_MAX);
    uint32_t k;
    memcpy(&k, &f, 4);
    return k < inf ? k : inf;
  };
  auto relax = [&](uint32_t label, float wgt) -> uint32_t {
    if (!weighted) return label + 1u;
    float a;
    memcpy(&a, &label, 4);
    const float f = a + wgt;
    uint32_t k;
    memcpy(&k, &f, 4);
    return k;
  };
  std::vector<uint32_t> dist((size_t)h.n_new, inf), expd((size_t)h.n_new, inf), bmin((size_t)h.nb, BLK_NONE), lab((size_t)NV);
  std::vector<uint32_t> chg((size_t)WORDS), pend((size_t)WORDS), fcur((size_t)WORDS), fnxt((size_t)WORDS);
  const int32_t s_new = h.perm[(size_t)src];
  dist[(size_t)s_new] = 0u;
  bmin[(size_t)(s_new / NV)] = 0u;
  uint32_t hi = key_add(0u, delta_bits);
  grx_block_stats_t st{};
  st.buckets = 1; st.blocks = h.nb; st.block_vertices = NV; st.cross_edges = h.n_cross;
  std::vector<int32_t> queue;
  for (;;) {
    uint32_t mn = BLK_NONE;
    queue.clear();
    for (int b = 0; b < h.nb; ++b) { mn = std::min(mn, bmin[(size_t)b]); if (bmin[(size_t)b] < hi) queue.push_back(b); }
    if (queue.empty()) {
      if (mn == BLK_NONE) break;
      hi = key_add(mn, delta_bits);
      if (hi <= mn) hi = mn + 1u;
      ++st.buckets;
      for (int b = 0; b < h.nb; ++b) if (bmin[(size_t)b] < hi) queue.push_back(b);
    }
    ++st.supersteps;
    st.activations += (int64_t)queue.size();
    for (int b : queue) {
      const size_t base = (size_t)b * NV;
      bmin[(size_t)b] = BLK_NONE;
      const unsigned short* off = &h.off[(size_t)b * OFFS];
      const unsigned short* tgt = &h.tgt[(size_t)b * NE];
      const float* wt = weighted ? &h.wt[(size_t)b * NE] : nullptr;
      std::fill(chg.begin(), chg.end(), 0u); std::fill(pend.begin(), pend.end(), 0u);
      std::fill(fcur.begin(), fcur.end(), 0u); std::fill(fnxt.begin(), fnxt.end(), 0u);
      bool any = false;
      for (int v = 0; v < NV; ++v) {
        lab[(size_t)v] = dist[base + v];
        if (dist[base + v] < expd[base + v]) {
          pend[(size_t)(v >> 5)] |= 1u << (v & 31);
          if (dist[base + v] < hi) { fcur[(size_t)(v >> 5)] |= 1u << (v & 31); chg[(size_t)(v >> 5)] |= 1u << (v & 31); any = true; }
        }
      }
      while (any) {
        any = false;
        for (int wd = 0; wd < WORDS; ++wd) {
          uint32_t wv = fcur[(size_t)wd];
          fcur[(size_t)wd] = 0u;
          while (wv) {
            const int v = wd * 32 + __builtin_ctz(wv);
            wv &= wv - 1u;
            const uint32_t lu = lab[(size_t)v];
            st.edges_relaxed += off[v + 1] - off[v];
            for (int e = off[v]; e < off[v + 1]; ++e) {
              const int t = tgt[e];
              const uint32_t cand = relax(lu, wt ? wt[e] : 1.0f);
              if (cand < lab[(size_t)t]) {
                lab[(size_t)t] = cand;
                chg[(size_t)(t >> 5)] |= 1u << (t & 31);
                if (cand < hi) { fnxt[(size_t)(t >> 5)] |= 1u << (t & 31); any = true; }
              }
            }
          }
        }
        fcur.swap(fnxt);
      }
      uint32_t left = BLK_NONE;
      for (int v = 0; v < NV; ++v) {
        const uint32_t bit = 1u << (v & 31);
        const bool c = (chg[(size_t)(v >> 5)] & bit) != 0u, p = (pend[(size_t)(v >> 5)] & bit) != 0u;
        if (!c && !p) continue;
        const uint32_t L = lab[(size_t)v];
        const size_t vg = base + v;
        if (c) {
          if (L < hi) {
            expd[vg] = L;
            st.edges_relaxed += h.xro[vg + 1] - h.xro[vg];
            for (int e = h.xro[vg]; e < h.xro[vg + 1]; ++e) {
              const int t = h.xci[(size_t)e];
              const uint32_t cand = relax(L, weighted ? h.xw[(size_t)e] : 1.0f);
              if (cand < dist[(size_t)t]) {
                // (an in-edge from another block: the target must carry the boundary mark the device relies on)
                if (!((h.bnd[(size_t)(t / NV) * WORDS + ((t % NV) >> 5)] >> (t & 31)) & 1u))
                  return fail(GRX_ERROR_INVALID_ARGUMENT, "boundary mark missing");
                dist[(size_t)t] = cand;
                bmin[(size_t)(t / NV)] = std::min(bmin[(size_t)(t / NV)], cand);
              }
            }
          }
          dist[vg] = std::min(dist[vg], L);
        }
        if (L >= hi) left = std::min(left, L);
      }
      bmin[(size_t)b] = std::min(bmin[(size_t)b], left);
    }
  }
  for (int32_t v = 0; v < csr->V; ++v) out_keys[v] = dist[(size_t)h.perm[(size_t)v]];
  if (stats) *stats = st;
  return GRX_SUCCESS;
}

// Is the block-asynchronous path to be used for this graph / kind?  Builds the structure on first use.
grx_status_t grx::blk_prepare(grx_context_t ctx, grx_graph_t g, bool weighted, bool* usable) {
  *usable = false;
  const int k = weighted ? 1 : 0;
  if (blk_env("GRX_BLOCK", GRX_BLOCK_DEFAULT) == 0) return GRX_SUCCESS;
  std::lock_guard<std::recursive_mutex> lk(g->prep_mu);
  if (g->blk_state[k] == 2) return GRX_SUCCESS;
  if (g->blk_state[k] == 1) { *usable = true; return GRX_SUCCESS; }
  lazy_state state(&g->blk_state[k]);  // an error return (a bad GRX_BLOCK_NV, a failed allocation) leaves it at 0: retried
  // road-like: few edges per vertex, and big enough that the level-synchronous search is thousands of launches deep
  if (g->V < blk_env("GRX_BLOCK_MIN_V", 1 << 16) || (long long)g->E >= 4ll * g->V || g->E <= 0) return state.done(2);
  if (weighted && (!g->w || g->weight_sum < 0.0 || !(g->weight_min >= 0.0f))) return state.done(2);
  const int nv = weighted ? blk_env("GRX_BLOCK_NV_W", 2048) : blk_env("GRX_BLOCK_NV", 4096);
  if (nv != 2048 && nv != 4096 && nv != 8192) return fail(GRX_ERROR_INVALID_ARGUMENT, "GRX_BLOCK_NV: 2048, 4096 or 8192");
  if (weighted && nv == 8192) return fail(GRX_ERROR_INVALID_ARGUMENT, "GRX_BLOCK_NV_W: 2048 or 4096");
  blk_graph* bg = nullptr;
  bool ok = false;
  grx_status_t st = blk_build(ctx, g, weighted, nv, 3 * nv, &bg, &ok);
  if (st != GRX_SUCCESS) return st;
  if (!ok) return state.done(2);
  g->blk[k] = bg;
  *usable = true;
  return state.done(1);
}

// One search.  weighted: float labels (d_out: float[V], FLT_MAX unreached), else depths (d_out: int32[V], INT_MAX).
grx_status_t grx::blk_search(grx_context_t ctx, grx_graph_t g, int32_t src, const grx_options_t& opt, bool weighted, void* d_out,
                             float* elapsed_ms) {
  blk_graph* bg = reinterpret_cast<blk_graph*>(g->blk[weighted ? 1 : 0]);
  if (!bg) return fail(GRX_ERROR_INVALID_ARGUMENT, "blk_search: no block structure");
  hipStream_t s = ctx->stream;
  GRX_HIP(ctx->blk_buf[0].reserve((size_t)bg->n_new * sizeof(uint32_t)));
  GRX_HIP(ctx->blk_buf[1].reserve((size_t)bg->n_new * sizeof(uint32_t)));
  GRX_HIP(ctx->blk_buf[2].reserve(((size_t)bg->nb * 2 + 64) * sizeof(uint32_t)));
  blk_run r{};
  r.dist = ctx->blk_buf[0].as<uint32_t>();
  r.expd = ctx->blk_buf[1].as<uint32_t>();
  r.bmin = ctx->blk_buf[2].as<uint32_t>();
  r.queue = reinterpret_cast<int32_t*>(r.bmin + bg->nb + 16);
  r.ctrl = ctx->d_ctrl;
  r.mailbox = ctx->d_mailbox;
  r.inf = weighted ? 0x7f7fffffu : (uint32_t)INT_MAX;
  if (weighted) {
    // bucket width: GRX_BLOCK_DELTA_W (default 128) mean edge weights -- about the weighted diameter of a block
    const double mean_w = g->E > 0 ? g->weight_sum / (double)g->E : 1.0;
    float d = (float)(std::max(1e-30, mean_w) * (double)blk_env("GRX_BLOCK_DELTA_W", 128));
    if (!(d > 0.0f) || !std::isfinite(d)) d = 1.0f;
    memcpy(&r.delta, &d, sizeof(float));
  } else {
    r.delta = (uint32_t)std::max(1, blk_env("GRX_BLOCK_DELTA", 256));  // hops
  }
  blk_dev d{};
  d.nb = bg->nb;
  d.nv_shift = bg->nv == 8192 ? 13 : (bg->nv == 4096 ? 12 : 11);
  d.n_new = bg->n_new;
  d.inv = bg->inv; d.off = bg->off; d.tgt = bg->tgt; d.wt = bg->wt; d.nedge = bg->nedge;
  d.xro = bg->xro; d.xci = bg->xci; d.xw = bg->xw; d.bnd = bg->bnd; d.xout = bg->xout;

  // problem.reset(), outside the timed region like the reference's
  hipLaunchKernelGGL(blk_reset_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, s, r, (int64_t)bg->n_new, bg->nb);
  ctx->h_mailbox[0] = 0;
  GRX_HIP(hipEventRecord(ctx->ev_begin, s));
  hipLaunchKernelGGL(blk_seed_kernel, dim3(1), dim3(64), 0, s, r, d, bg->perm, src, weighted ? 1 : 0);
  // resident workgroups: what the kernel's LDS allows per CU
  const int per_cu = std::max(1, std::min(8, (160 * 1024) / (bg->nv * 4 + (bg->nv + 8) * 2 + bg->ne * (weighted ? 6 : 2) + bg->nv / 32 * 20 + 256)));
  const int grid = ctx->num_cus * std::max(1, std::min(per_cu, blk_env("GRX_BLOCK_WG_PER_CU", per_cu)));
  const bool profile = (opt.engine_flags & GRX_FLAG_PROFILE) != 0;
  ctx->levels.clear();
  hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
  if (profile) for (auto& ev : pe) GRX_HIP(hipEventCreate(&ev));
  hipError_t launch_err = hipSuccess;
  int64_t prof_e = 0, prof_a = 0;
  grx_status_t st = run_levels(ctx, opt, [&](hipStream_t stream, int) {
    if (profile) (void)hipEventRecord(pe[0], stream);
    hipLaunchKernelGGL(blk_head_kernel, dim3(1), dim3(1024), 0, stream, r, d, weighted ? 1 : 0);
    if (profile) (void)hipEventRecord(pe[1], stream);
    if (weighted) {
      if (bg->nv == 2048) hipLaunchKernelGGL((blk_kernel<2048, 6144, true>), dim3(grid), dim3(64), 0, stream, r, d);
      else hipLaunchKernelGGL((blk_kernel<4096, 12288, true>), dim3(grid), dim3(128), 0, stream, r, d);
    } else {
      if (bg->nv == 2048) hipLaunchKernelGGL((blk_kernel<2048, 6144, false>), dim3(grid), dim3(64), 0, stream, r, d);
      else if (bg->nv == 4096) hipLaunchKernelGGL((blk_kernel<4096, 12288, false>), dim3(grid), dim3(128), 0, stream, r, d);
      else hipLaunchKernelGGL((blk_kernel<8192, 24576, false>), dim3(grid), dim3(256), 0, stream, r, d);
    }
    if (profile) {
      (void)hipEventRecord(pe[2], stream);
      (void)hipEventSynchronize(pe[2]);
      level_rec rec{};
      (void)hipEventElapsedTime(&rec.other_ms, pe[0], pe[1]);
      (void)hipEventElapsedTime(&rec.advance_ms, pe[1], pe[2]);
      ctx->levels.push_back(rec);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) launch_err = e;
  }, [&](const ctrl_t& h) {
    if (!profile || ctx->levels.empty()) return;
    level_rec& rec = ctx->levels.back();
    rec.frontier_size = h.vertices_visited - prof_a;  // block activations of the superstep
    rec.edges = h.edges_visited - prof_e;             // edges relaxed in it
    rec.bottom_up = 4;                                // body: block-asynchronous superstep
    prof_a = h.vertices_visited;
    prof_e = h.edges_visited;
    if (h.done && rec.frontier_size == 0 && rec.edges == 0) ctx->levels.pop_back();
  }, /*first_batch=*/16);
  if (profile) for (auto& ev : pe) (void)hipEventDestroy(ev);
  if (st != GRX_SUCCESS) return st;
  if (launch_err != hipSuccess) return fail(GRX_ERROR_HIP, hipGetErrorString(launch_err));
  hipLaunchKernelGGL(blk_final_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, s, r, d, g->ro, reinterpret_cast<uint32_t*>(d_out));
  GRX_HIP(hipEventRecord(ctx->ev_end, s));
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, s));
  GRX_HIP(hipEventSynchronize(ctx->ev_end));
  GRX_HIP(hipStreamSynchronize(s));
  float ms = 0;
  GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  const ctrl_t& h = *ctx->h_ctrl;
  ctx->block_stats.edges_relaxed = h.edges_visited;
  ctx->block_stats.activations = h.vertices_visited;
  ctx->block_stats.supersteps = h.level;
  ctx->block_stats.buckets = h.spare[2];
  ctx->block_stats.blocks = bg->nb;
  ctx->block_stats.block_vertices = bg->nv;
  ctx->block_stats.cross_edges = bg->n_cross;
  ctx->block_stats.build_ms = bg->host_ms;
  // BFS: the reference's counters (out-edges of the reached vertices, each once; depth = deepest level + 1);
  // weighted SSSP: edges relaxed (re-relaxations included), as the near-far schedule reports them
  ctx->stats.edges_visited = weighted ? h.edges_visited : h.bu_probes;
  ctx->stats.vertices_visited = h.bu_open;
  ctx->stats.search_depth = weighted ? h.level : (int32_t)((uint32_t)h.spare[3] + 1u);
  ctx->stats.elapsed_ms = ms;
  ctx->stats.n_levels_recorded = (int32_t)ctx->levels.size();
  ctx->stats.reserved = (float)h.level;
  if (elapsed_ms) *elapsed_ms = ms;
  return GRX_SUCCESS;
}
