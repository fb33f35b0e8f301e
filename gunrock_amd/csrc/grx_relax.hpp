// grx_relax.hpp -- BINNED RELAXATION: the fat levels of a weighted label-correcting SSSP on a dense graph as two streaming
// passes, the relaxation's minimum taken in LDS.
//
// What it replaces in the reference: the advance of sssp.hxx:116-130 -- nd = dist[src] + w; old = atomicMin(&dist[nbr], nd);
// keep if nd < old -- and the filter behind it (:132-151), for the levels where that pair is bound by its random accesses.
//
// Why.  The relax-per-edge advance (advance_block with sssp_policy) probes the label of every neighbour and follows up with a
// device-scope atomic where the probe says "may improve": on a multi-XCD part that atomic executes at the memory side
// (~20 G/s for the whole device, grx_bin.hpp) and the probe moves a 64-byte sector per edge.  Round 4, LJ stand-in with
// U{1..1000} weights: 228 M relaxations in 6.1 ms = 37 G relaxations/s, 0.13 of the HBM roofline of SURVEY 8d's 28 B per edge --
// while the BFS of the same graph runs its fat levels at 220-330 G edges/s on the binned kernels.
//
// Here a fat level of the plain (label-correcting) schedule runs as
//   1. SCATTER (bin_scatter2_block<.., VAL = true>, grx_bin.hpp): the frontier is expanded exactly as for the BFS, and for
//      every edge the pair (offset of the target inside its bin: 16 bits, fl(label of the source + weight): 32 bits) is
//      appended to the BIN of the target's vertex range.  Bins are <= 16384 vertices wide and there are up to 1024 of them
//      (cut once per graph by in-edges, capacities static, as for the BFS).
//   2. SWEEP (relax_sweep_block, below): a workgroup of 1024 threads takes one bin -- or one part of a bin that received
//      more than its share of the level's entries (hub ranges) --, copies the labels of the bin's vertex range into LDS
//      (<= 64 KB), streams the bin's entries through them with ds_min_u32 (non-negative floats order like their bit
//      patterns), and writes back what changed: plain coalesced stores when the bin has one part (nobody else touches those
//      labels in this launch), one device-scope atomicMin per CHANGED VERTEX and part otherwise (+ the per-level stamp that
//      decides which part emits the vertex).  The changed vertices leave in ascending order as tiles of the next frontier,
//      with their entries of the chunk map (sweep2_emit, shared with the BFS sweep).
// No global atomic and no random global access per edge.  The fixed point of a label-correcting search does not depend on the
// relaxation schedule (grx_sssp.hip), so distances stay bit-identical to the oracle's.
#pragma once

#include "grx_bin.hpp"

namespace grx {

constexpr int RB_BLOCK = 1024;
constexpr int RB_SHIFT = 14;                    // widest bin: 16384 vertices = 64 KB of labels in LDS
constexpr int RB_WIDTH = 1 << RB_SHIFT;
constexpr int RB_MAX_BINS = BIN_MAX * SC2_SUB;  // 1024: one histogram counter per thread of the scatter
constexpr int RB_SEG_WORDS = RB_BLOCK / 4;      // words of the changed bitmap per expansion round (a thread expands one byte)
constexpr int RB_PART_MIN = 1 << 15;            // a bin is cut into parts of at least this many entries
constexpr int RB_U = 2;                         // 16-byte offset loads (8 entries each) per thread and round
#ifndef GRX_RB_RING
#define GRX_RB_RING 2
#endif
constexpr int RB_RING = GRX_RB_RING;            // rounds of entry loads in flight per thread (3 x RB_U 16-byte loads each)
constexpr int RB_QUEUE_SLOT = 15;               // bin_args::queue word (x BIN_PAD) the sweep draws its items from

// (Measured and removed, round 4 call 17: a software-pipelined build of the scatter -- owner map and column / weight loads of batch
// i + 1 issued ahead of the reservation, sort and copy-out of batch i, same phases and barriers, +25 VGPRs -- was correct and 2-3 %
// SLOWER on the LJ and kron stand-ins: profiles/history/r4_ab_relax_scatter_pipelined_rejected.txt.  The scatter is
// bin_scatter2_block<.., VAL = true> of grx_bin.hpp.)

struct relax_sweep_smem {
  static constexpr int LIST = RB_SEG_WORDS * 32 + TILE;
  static constexpr int MAX_TILES = LIST / TILE + 1;
  unsigned d[RB_WIDTH];           // labels of the bin's vertex range, ordered bit patterns
  unsigned chg[RB_WIDTH / 32];    // vertices of the range this workgroup emits
  int list[LIST];
  int pre[RB_MAX_BINS + 1];
  int fillv[RB_MAX_BINS];
  int wave[RB_BLOCK / 64 + 1];
  int sum[MAX_TILES][4];
  int ttot[64];
  int cpre[64];
  int tile_base;
  int chunk_base;
  int n_chunks;
  int item;
};

__device__ __forceinline__ void relax_sweep_block(const pipe_args& a, const bin_args& bn, ctrl_t* c, int level,
                                                  relax_sweep_smem& sm, int p) {
  using S = relax_sweep_smem;
  constexpr int NT = RB_BLOCK;
  constexpr int EPL = 8;  // entries per 16-byte load of offsets
  const int tid0 = threadIdx.x;
  int tid = tid0;
  const int q = p ^ 1;
  if (blockIdx.x == 0 && tid == 0) c->map_level = level + 1;  // the chunk map and the counters of the next level come from this kernel
  int fill = 0;
  if (tid < bn.nb) fill = bn.fill[(unsigned)(tid * BIN_PAD)];
  int tot_fill;
  (void)dev::block_exclusive_sum<NT>(fill, sm.wave, &tot_fill);
  // every out-edge of the level's frontier is one entry (see bin_sweep2_block: the same safety net for the scatter's queues)
  if ((long long)tot_fill != c->q_edges[p]) {
    if (blockIdx.x == 0 && tid == 0) {
      c->mid_err = 2;
      c->done = 1;
      a.mailbox[10] = 2;
      __threadfence_system();
      a.mailbox[0] = 1;
    }
    return;
  }
  const int parts = max(1, bn.sweep_items - bn.nb);
  const int PART = max(RB_PART_MIN, ((tot_fill / parts) + EPL) & ~(EPL - 1));
  int tot_items;
  const int ex0 = dev::block_exclusive_sum<NT>((fill + PART - 1) / PART, sm.wave, &tot_items);
  sm.pre[tid] = ex0;
  sm.fillv[tid] = fill;
  int* qhead = &bn.queue[(unsigned)(RB_QUEUE_SLOT * BIN_PAD)];  // zeroed by the head kernel with the fill counters
  if (tid == 0) {
    sm.pre[RB_MAX_BINS] = tot_items;
    sm.item = atomicAdd(qhead, 1);
  }
  __syncthreads();
  unsigned* dist_u = reinterpret_cast<unsigned*>(bn.rdist);
  const int4* off4 = reinterpret_cast<const int4*>(bn.bins);
  const int4* val4 = reinterpret_cast<const int4*>(bn.rval);
  int n_list = 0;  // uniform: vertices waiting in sm.list
  auto emit_list = [&](bool all) {
    const int k = all ? (n_list + TILE - 1) / TILE : n_list / TILE;
    const int n_emit = all ? n_list : k * TILE;
    sweep2_emit<NT>(a, c, q, sm, n_emit);
    const int rem = n_list - n_emit;
    int keep = 0;
    if (tid < rem) keep = sm.list[n_emit + tid];
    __syncthreads();
    if (tid < rem) sm.list[tid] = keep;
    n_list = rem;
    __syncthreads();
  };
  for (;;) {
    const int item = __builtin_amdgcn_readfirstlane(sm.item);
    if (item >= tot_items) break;
    __syncthreads();  // everybody has read the item
    tid = tid0;
    asm volatile("" : "+v"(tid));  // (per-thread constants are re-derived per item instead of living in VGPRs)
    // the next item's ticket: its round trip overlaps this item.  The address is made opaque (see bin_scatter2_block): on a
    // uniform address the compiler's lowering of a one-lane atomic waits for the result on the spot, in front of this item's
    // first loads.
    int next_item = 0;
    {
      int vz;
      asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
      if (tid == 0) next_item = atomicAdd(qhead + vz, 1);
    }
    int b = 0;  // largest b with pre[b] <= item (bins without items are skipped over)
#pragma unroll
    for (int step = RB_MAX_BINS / 2; step >= 1; step >>= 1)
      if (sm.pre[b + step] <= item) b += step;
    const int fb = sm.fillv[b];
    const int e0 = (item - sm.pre[b]) * PART;
    const int n_e = min(fb, e0 + PART) - e0;
    const bool single = fb <= PART;  // uniform: this workgroup is the only writer of the range's labels in this launch
    const int lo = bn.off[b] + e0, hi = lo + n_e;
    const int vbase = bn.v0[b];
    const int nv = bn.v0[b + 1] - vbase;  // a multiple of the granule (>= 1024 vertices); the last bin may reach past V
    const int i4_first = lo / EPL, i4_last = (hi - 1) / EPL;
    // a ring of RB_RING rounds of entry loads in flight, the buffers indexed by the constants of an unrolled group: see
    // bin_sweep2_block (the round-4 loop `cur = next; LOAD(r + 1, next); process(cur)` waited out every round's full latency)
    int4 rg_o[RB_RING][RB_U], rg_v[RB_RING][RB_U][2];
    auto LOAD = [&](int r, int4(&o)[RB_U], int4(&v)[RB_U][2]) {
#pragma unroll
      for (int u = 0; u < RB_U; ++u) {
        int idx = i4_first + (r * RB_U + u) * NT + tid;
        idx = idx < i4_last ? idx : i4_last;
        o[u] = off4[idx];
        v[u][0] = val4[2 * (size_t)idx];
        v[u][1] = val4[2 * (size_t)idx + 1];
      }
    };
#pragma unroll
    for (int k = 0; k < RB_RING; ++k) LOAD(k, rg_o[k], rg_v[k]);  // the first entries are on their way while the labels are copied
    {
      // (all loads of the slice issued before the first LDS store: a loop of load -> store pairs is one memory round trip
      // per iteration, 16 of them per item -- measured: ~125 us of every sweep, whatever the level's size)
      unsigned g16[RB_WIDTH / NT];
#pragma unroll
      for (int k = 0; k < RB_WIDTH / NT; ++k) {
        const int i = k * NT + tid;
        g16[k] = (i < nv && (vbase + i) < a.V) ? dist_u[vbase + i] : 0u;
      }
#pragma unroll
      for (int k = 0; k < RB_WIDTH / NT; ++k)
        if (k * NT < nv) sm.d[k * NT + tid] = g16[k];
    }
    __syncthreads();
    // A. entries -> minimum in LDS
    auto PROCESS = [&](int r, const int4(&co)[RB_U], const int4(&cv)[RB_U][2]) {
#pragma unroll
      for (int u = 0; u < RB_U; ++u) {
        const int idx = i4_first + (r * RB_U + u) * NT + tid;
        const int g0 = idx * EPL;
        const unsigned q4[4] = {(unsigned)co[u].x, (unsigned)co[u].y, (unsigned)co[u].z, (unsigned)co[u].w};
        const unsigned x8[EPL] = {(unsigned)cv[u][0].x, (unsigned)cv[u][0].y, (unsigned)cv[u][0].z, (unsigned)cv[u][0].w,
                                  (unsigned)cv[u][1].x, (unsigned)cv[u][1].y, (unsigned)cv[u][1].z, (unsigned)cv[u][1].w};
        unsigned o8[EPL], cur[EPL];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o8[2 * j] = q4[j] & 0xffffu;
          o8[2 * j + 1] = q4[j] >> 16;
        }
        // (unconditional LDS reads from clamped positions first -- a label hit by many lanes at once is a broadcast read
        // where an atomic on one word serialises -- then the atomics that can still lower something)
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          const int gi = g0 + j;
          const bool ok = idx <= i4_last && gi >= lo && gi < hi;
          o8[j] = ok ? (o8[j] & (unsigned)(RB_WIDTH - 1)) : 0u;
          cur[j] = ok ? sm.d[o8[j]] : 0u;  // 0: nothing is below it
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j)
          if (x8[j] < cur[j]) (void)__hip_atomic_fetch_min(&sm.d[o8[j]], x8[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    };
    const int rounds = (i4_last - i4_first + RB_U * NT) / (RB_U * NT);
    for (int r0 = 0; r0 < rounds; r0 += RB_RING) {
#pragma unroll
      for (int k = 0; k < RB_RING; ++k) {
        PROCESS(r0 + k, rg_o[k], rg_v[k]);   // (rounds past the end find every entry out of range)
        LOAD(r0 + k + RB_RING, rg_o[k], rg_v[k]);
      }
    }
    __syncthreads();
    // B. what changed goes back (vertex order, lanes on consecutive labels); the ballots are the changed bitmap.  The current
    // labels are re-read in one batch (a part of a bin may have been overtaken by another part meanwhile).
    {
      constexpr int K = RB_WIDTH / NT;
      unsigned g16[K], l16[K];
      bool mine[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int i = k * NT + tid;
        g16[k] = (i < nv && (vbase + i) < a.V) ? dist_u[vbase + i] : 0u;
      }
#pragma unroll
      for (int k = 0; k < K; ++k) l16[k] = sm.d[k * NT + tid];
#pragma unroll
      for (int k = 0; k < K; ++k) mine[k] = k * NT < nv && l16[k] < g16[k];
      if (single) {
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (mine[k]) dist_u[vbase + k * NT + tid] = l16[k];
      } else {
        // parts of one bin race here, and only here: one device-scope atomic per changed vertex and part, then the stamp
        // of the level decides which part emits the vertex
#pragma unroll
        for (int k = 0; k < K; ++k) g16[k] = mine[k] ? atomicMin(&dist_u[vbase + k * NT + tid], l16[k]) : 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) mine[k] = mine[k] && l16[k] < g16[k];
        int st[K];
#pragma unroll
        for (int k = 0; k < K; ++k) st[k] = mine[k] ? atomicExch(&bn.rstamp[vbase + k * NT + tid], level) : level;
#pragma unroll
        for (int k = 0; k < K; ++k) mine[k] = mine[k] && st[k] != level;
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (k * NT < nv) {  // uniform
          const unsigned long long m = dev::ballot(mine[k]);
          if ((tid & 63) == 0) {
            const int i = k * NT + tid;
            sm.chg[i >> 5] = (unsigned)m;
            sm.chg[(i >> 5) + 1] = (unsigned)(m >> 32);
          }
        }
      }
    }
    __syncthreads();
    // C. changed bits -> ascending vertex ids -> tiles of the next frontier
    const int words = nv >> 5;
    for (int s0 = 0; s0 < words; s0 += RB_SEG_WORDS) {
      const int w = s0 + (tid >> 2);
      unsigned byte = w < words ? (sm.chg[w] >> ((tid & 3) * 8)) & 0xffu : 0u;
      int tot;
      const int ex = dev::block_exclusive_sum<NT>(__popc(byte), sm.wave, &tot);
      if (tot == 0) continue;
      if (n_list + tot > S::LIST) emit_list(false);  // n_list >= TILE here: tot <= LIST - TILE
      int pos = n_list + ex;
      const int v_first = vbase + (w << 5) + (tid & 3) * 8;
      while (byte) {
        sm.list[pos++] = v_first + __ffs(byte) - 1;
        byte &= byte - 1u;
      }
      n_list += tot;
      __syncthreads();
    }
    // (no emission at the end of an item: the list is emitted when the next segment would not fit, and once at the end --
    // an emission is a chain of ~5 dependent round trips whatever its size, and a workgroup takes 2-3 items per level)
    if (tid == 0) sm.item = next_item;
    __syncthreads();
  }
  if (n_list > 0) emit_list(true);
}

}  // namespace grx
