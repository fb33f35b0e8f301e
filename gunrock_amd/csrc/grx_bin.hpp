// grx_bin.hpp -- BINNED top-down advance: the fat levels of a forward (top-down only) BFS as two
// streaming passes instead of one random probe per edge.
//
// What it replaces in the reference: the same merge-path advance + filter pair as
// grx_frontier.hpp (operators/advance/merge_path.hxx:112-362, filter/predicated.hxx:12-39),
// for the levels where that pair is bound by random label probes.
//
// Why.  The claim-per-edge advance (advance_block) reads the label of every neighbour: 31 M
// scattered 4-byte probes on the fat level of the LJ stand-in, each moving a 64-byte sector across
// the fabric (rocprofv3, round 1: 1.88 GB of L2-miss traffic for 0.41 GB of algorithmic bytes,
// L2 hit rate 35 %, 0.54 ms).  More than 90 % of those probes only learn "already visited".
// Here a fat level runs in two kernels whose global traffic is all coalesced streams:
//   1. SCATTER (bin_scatter_block, inside the level kernel): the frontier is expanded exactly as
//      in advance_block (tiles, 2048-edge chunks, lanes on consecutive edges), but a neighbour id
//      is not probed -- it is appended to the BIN of its vertex range (bin = id >> shift, <= 256
//      bins).  A workgroup sorts 4 chunks (8192 ids) by bin in LDS (histogram + rank with LDS
//      atomics, block scan) and writes each bin's run with ONE reservation atomic per bin and
//      batch; runs leave LDS as contiguous segments.  Bin capacities are STATIC: the number of
//      in-edges of the bin's vertex range (each edge is traversed at most once per level), so the
//      bins are one E-entry array that can never overflow and needs no size pass.
//   2. CLAIM (bin_claim_block, its own launch): a workgroup takes a slice of <= 8192 entries of
//      ONE bin, copies that bin's slice of the visited bitmap into LDS (<= 16 KB), and tests /
//      sets the bit of every entry THERE.  Only ids new to the workgroup go on to the
//      reference's atomicMin on the label (bfs.hxx:117-119) -- about one per discovered vertex
//      instead of one probe per edge -- and the winners are compacted into frontier tiles exactly
//      as advance_block does.
// The head kernel chooses per level (degree sum of the frontier >= bin_args::min_edges); all other
// levels of the run go through advance_block, whose discoveries keep the same visited bitmap
// current (bfs_policy::on_accept), so the formats never need converting.
#pragma once

#include "grx_bfs_kernels.hpp"

namespace grx {

constexpr int BIN_MAX = ADV_BLOCK;   // bins: one thread of a workgroup per bin
constexpr int BIN_BATCH = 4;         // chunks sorted together (one reservation atomic per bin and batch:
                                     // a single word sustains only ~90 atomics/us)
constexpr int BIN_SLICE = 8192;      // entries per claim work item
constexpr int BIN_PAD = 32;          // ints between two fill counters (each in its own 128-byte line)
constexpr int BIN_SHIFT_MAX = 17;    // widest vertex range per bin: 131072 vertices = 16 KB of bitmap in LDS

struct bin_args {
  int32_t* bins;          // E entries; bin b owns [off[b], off[b + 1])
  const int32_t* off;     // nb + 1 static offsets
  int32_t* fill;          // entries in bin b this level at fill[b * BIN_PAD] (zeroed by the head kernel)
  int32_t shift;          // bin of vertex n = n >> shift
  int32_t nb;             // bins in use (<= BIN_MAX)
  long long min_edges;    // a level with at least this many out-edges is binned (0: never)
  unsigned* visited;      // V-bit visited bitmap of the run
  int32_t visited_words;
  int32_t* dist;
};

struct bin_scatter_smem {
  int seg[TILE + 1];
  int start[TILE];
  int wave[ADV_BLOCK / 64 + 1];
  int hist[BIN_MAX];
  int off[BIN_MAX];
  int delta[BIN_MAX];
  int sorted[BIN_BATCH * CHUNK];
};

// Phase 1.  Chunks blockIdx.x, + gridDim.x, ... of the frontier with parity p, BIN_BATCH at a time.
__device__ __forceinline__ void bin_scatter_block(const pipe_args& a, const bin_args& bn, bin_scatter_smem& sm, int p,
                                                  int total_chunks, const int* chunk_tile) {
  static_assert(BIN_MAX == ADV_BLOCK, "one thread per bin");
  const int tid = threadIdx.x;
  const int32_t* in = a.frontier[p];
  const int stride = (int)gridDim.x;
  const int shift = bn.shift;
  for (int u0 = (int)blockIdx.x; u0 < total_chunks; u0 += stride * BIN_BATCH) {
    sm.hist[tid] = 0;
    __syncthreads();
    int n_k[BIN_BATCH][ADV_ITEMS], r_k[BIN_BATCH][ADV_ITEMS];
#pragma unroll
    for (int j = 0; j < BIN_BATCH; ++j) {
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        n_k[j][k] = 0;
        r_k[j][k] = -1;
      }
      const int unit = u0 + j * stride;
      if (unit < total_chunks) {  // uniform over the workgroup
        const int2 tl = reinterpret_cast<const int2*>(chunk_tile)[unit];
        const int v = in[(size_t)tl.x * TILE + tid];
        int rs = 0, deg = 0;
        if (v >= 0) {
          rs = a.ro[v];
          deg = a.ro[v + 1] - rs;
        }
        int tot;
        const int ex = dev::block_exclusive_sum<ADV_BLOCK>(deg, sm.wave, &tot);
        sm.seg[tid] = ex;
        sm.start[tid] = rs;
        if (tid == 0) sm.seg[TILE] = tot;
        __syncthreads();
        const int a0 = tl.y * CHUNK;
        const int a_end = min(tot, a0 + CHUNK);
        int e_k[ADV_ITEMS];
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) {
          const int atom = a0 + k * ADV_BLOCK + tid;
          e_k[k] = -1;
          if (atom < a_end) {
            int lo = 0;
#pragma unroll
            for (int step = TILE / 2; step >= 1; step >>= 1)
              if (sm.seg[lo + step] <= atom) lo += step;
            e_k[k] = sm.start[lo] + (atom - sm.seg[lo]);
          }
        }
        // all column-index loads of the chunk in flight together (lanes past the end read edge 0)
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) n_k[j][k] = a.ci[e_k[k] >= 0 ? e_k[k] : 0];
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k)
          if (e_k[k] >= 0) r_k[j][k] = atomicAdd(&sm.hist[n_k[j][k] >> shift], 1);
        __syncthreads();  // seg / start are rewritten by the next chunk
      }
    }
    // one reservation per non-empty bin, issued ahead of the scan so that its round trip overlaps
    const int cnt = sm.hist[tid];
    int gbase = 0;
    if (cnt > 0) gbase = atomicAdd(&bn.fill[tid * BIN_PAD], cnt);
    const int boff = tid < bn.nb ? bn.off[tid] : 0;
    int tot;
    const int ex = dev::block_exclusive_sum<ADV_BLOCK>(cnt, sm.wave, &tot);
    sm.off[tid] = ex;
    sm.delta[tid] = boff + gbase - ex;  // global slot of sorted position i of this bin: delta + i
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BIN_BATCH; ++j)
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k)
        if (r_k[j][k] >= 0) sm.sorted[sm.off[n_k[j][k] >> shift] + r_k[j][k]] = n_k[j][k];
    __syncthreads();
    for (int i = tid; i < tot; i += ADV_BLOCK) {
      const int n = sm.sorted[i];
      bn.bins[(size_t)(sm.delta[n >> shift] + i)] = n;
    }
    __syncthreads();
  }
}

struct bin_claim_smem {
  unsigned bm[1 << (BIN_SHIFT_MAX - 5)];
  int pre[BIN_MAX + 1];
  int fillv[BIN_MAX];
  int out[TILE + CHUNK];
  int wave[ADV_BLOCK / 64 + 1];
  int cnt;
  int res[3];
  emit_smem emit;
};

// Phase 2.  Discoveries are emitted as tiles of parity p ^ 1; pol.on_accept keeps the global visited
// bitmap current; pol.next_depth is the depth being assigned.
__device__ __forceinline__ void bin_claim_block(const pipe_args& a, const bin_args& bn, ctrl_t* c, bfs_policy& pol,
                                                bin_claim_smem& sm, int p) {
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  if (tid == 0) { sm.cnt = 0; sm.res[0] = 0; sm.res[1] = 0; }
  // work items: bin b contributes ceil(fill[b] / BIN_SLICE) slices
  int fill = 0;
  if (tid < bn.nb) fill = bn.fill[tid * BIN_PAD];
  int tot_items;
  const int ex = dev::block_exclusive_sum<ADV_BLOCK>((fill + BIN_SLICE - 1) / BIN_SLICE, sm.wave, &tot_items);
  sm.pre[tid] = ex;
  sm.fillv[tid] = fill;
  if (tid == 0) sm.pre[BIN_MAX] = tot_items;
  __syncthreads();
  const int depth = pol.next_depth;
  const int words = 1 << (bn.shift - 5);
  for (int item = (int)blockIdx.x; item < tot_items; item += (int)gridDim.x) {
    int b = 0;  // largest b with pre[b] <= item: the bin the item belongs to (empty bins are skipped over)
#pragma unroll
    for (int step = BIN_MAX / 2; step >= 1; step >>= 1)
      if (sm.pre[b + step] <= item) b += step;
    const int e0 = (item - sm.pre[b]) * BIN_SLICE;
    const int n_e = min(sm.fillv[b], e0 + BIN_SLICE) - e0;
    const int32_t* src = bn.bins + (size_t)bn.off[b] + e0;
    const int vbase = b << bn.shift;
    // this bin's slice of the visited bitmap -> LDS (words past the end of the bitmap: all visited)
    for (int w = tid; w < words; w += ADV_BLOCK) {
      const int gw = (vbase >> 5) + w;
      sm.bm[w] = gw < bn.visited_words ? bn.visited[gw] : ~0u;
    }
    __syncthreads();
    for (int r0 = 0; r0 < n_e; r0 += CHUNK) {
      int n_k[ADV_ITEMS], old_k[ADV_ITEMS];
      bool cand_k[ADV_ITEMS];
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        const int i = r0 + k * ADV_BLOCK + tid;
        n_k[k] = src[i < n_e ? i : 0];  // n_e > 0: entry 0 exists
      }
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        const int i = r0 + k * ADV_BLOCK + tid;
        cand_k[k] = false;
        if (i < n_e) {
          const int local = n_k[k] - vbase;
          const unsigned bit = 1u << (local & 31);
          // plain read first: visited hubs are hit by many lanes at once, and a read broadcasts
          // where an atomic on one word would serialise
          if (!(sm.bm[local >> 5] & bit)) cand_k[k] = (atomicOr(&sm.bm[local >> 5], bit) & bit) == 0u;
        }
      }
      // the reference's claim for the ids new to this workgroup, all issued before any result is used
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        old_k[k] = 0;
        if (cand_k[k]) old_k[k] = atomicMin(&bn.dist[n_k[k]], depth);
      }
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        const bool keep = cand_k[k] && depth < old_k[k];
        const unsigned long long m = dev::ballot(keep);
        if (m) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&sm.cnt, __popcll(m));
          base = __shfl(base, 0, 64);
          if (keep) {
            sm.out[base + dev::mask_rank(m)] = n_k[k];
            pol.on_accept(n_k[k]);
          }
        }
      }
      __syncthreads();
      int cnt = sm.cnt;
      if (cnt >= TILE) {
        const int k = cnt / TILE;
        emit_full_tiles(a, c, p ^ 1, sm.out, cnt - k * TILE, k, sm.emit, sm.res);
        cnt -= k * TILE;
      }
      if (tid == 0) sm.cnt = cnt;
      __syncthreads();
    }
  }
  const int rem = sm.cnt;
  if (rem > 0) emit_tile(a, c, p ^ 1, sm.out, 0, rem, sm.wave, sm.res);
  __syncthreads();
  release_tiles(a, sm.res);
}

// Per-graph static part: how many in-edges fall into each bin's vertex range (= its capacity).
// One pass over the column indices, once per graph.  <<<grid, 256>>>
__global__ void bin_count_kernel(const int32_t* __restrict__ ci, int64_t E, int shift, int32_t* cnt) {
  __shared__ int s_hist[BIN_MAX];
  s_hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
    const unsigned b = (unsigned)ci[e] >> shift;  // an id outside [0, V) is left uncounted: the host sees the shortfall
    if (b < (unsigned)BIN_MAX) atomicAdd(&s_hist[b], 1);
  }
  __syncthreads();
  const int v = s_hist[threadIdx.x];
  if (v) atomicAdd(&cnt[threadIdx.x], v);
}

}  // namespace grx
