// grx_frontier.hpp -- the device-driven frontier pipeline shared by BFS and SSSP.
//
// What it replaces in the reference (paths relative to /root/reference):
//   framework/enactor.hxx:243-288            enact(): host while-loop, >= 2 blocking syncs/level
//   operators/advance/helpers.hxx:41-111     per-level rocThrust scan + 4-byte D2H
//   operators/advance/merge_path.hxx:112-362 merge-path advance (strided 11 atoms per thread)
//   operators/filter/predicated.hxx:12-39    copy_if compaction (+ size readback)
//
// MI355X design
//   * The frontier is a queue of TILEs (256 vertex slots, -1 = empty slot).  A
//     tile carries its degree sum; a level's work is cut into CHUNKs of 2048
//     edges ("atoms") that never straddle a tile, so a workgroup stages ONE tile
//     (vertex ids, row starts, block-scanned degrees) in LDS and finds the owner
//     of each atom by an 8-probe LDS binary search.  Lanes of a wave take
//     CONSECUTIVE atoms, so column-index reads are coalesced inside a row.
//   * plan_kernel (one workgroup) turns per-tile chunk counts into the
//     chunk -> tile map of the level; it also detects the empty frontier.
//   * advance_kernel runs the user policy per edge and compacts accepted
//     neighbours on the fly: wave ballot + mbcnt rank inside the wave, an LDS
//     counter per workgroup, and ONE global atomic per 256 emitted vertices
//     (an output tile), whose degree sum is computed at emission time so the
//     next level needs no separate scan over the frontier.
//   * No kernel takes a level-dependent argument: sizes, parity and level come
//     from a device control block, so the host enqueues levels blindly (or
//     replays a hipGraph) and only looks at a `done` flag.
#pragma once

#include <type_traits>

#include "grx_common.hpp"
#include <gunrock/hip/wave.hxx>

namespace grx {

constexpr int TILE = 256;            // frontier slots per tile
constexpr int ADV_BLOCK = 256;       // threads per advance workgroup
constexpr int ADV_ITEMS = 8;         // atoms per thread per chunk
constexpr int CHUNK = ADV_BLOCK * ADV_ITEMS;  // 2048 atoms
constexpr int PLAN_BLOCK = 1024;
constexpr int TILE_RESERVE = 4;      // tile indices a workgroup reserves per global atomic
#ifndef GRX_MID_WGS
#define GRX_MID_WGS 32
#endif
constexpr int MID_FLAG_WORDS = 2 * GRX_MID_WGS;   // grx_mid.hpp: 2 x MID_WGS barrier words

struct pipe_args {
  const int32_t* ro;
  const int32_t* ci;
  const float* w;
  int32_t V;
  ctrl_t* ctrl;
  int32_t* mailbox;       // host-pinned, device-visible
  int32_t* frontier[2];   // tiled queues by level parity
  int32_t* tile_chunks;   // chunks per tile of the CURRENT input frontier
  int32_t* tile_sums;     // degree sum per tile of the current input frontier
  int32_t* tile_count;    // valid vertices per tile
  int32_t* chunk_tile;    // per chunk: int2 {owning tile, chunk index inside the tile}
  const long long* bu_part;  // direction-optimising BFS: 4 words per bottom-up workgroup (word 0 >> 40 = its tiles)
  void* mid_aux;             // grx_mid.hpp: {row start, degree} of the first entries of the flat queues, per parity
  void* mid_aux2;            // grx_mid.hpp, second version: int4 {row start, degree, state, -} per entry of the private regions, per parity
  unsigned long long* mid_flags;  // second version: 2 x MID_WGS barrier / count words (zeroed by the head kernel that chooses mode 3)
  int32_t mid_version;       // 1 | 2
  int32_t mid_seg_cap;       // second version: entries of a private region in use (<= MID_SEG; tests shrink it to reach the overflow path)
  int32_t mid_exit_v;        // second version: a frontier beyond this many vertices goes back to the regular kernels (<= MID_EXIT_V)
  int32_t mid_exit_e;        // ... or beyond this many out-edges (MID_EXIT_E)
  int32_t mid_hub_deg;       // ... or averaging more than this many out-edges per vertex (0: no such rule)
  int32_t mid_refill_max;    // policies that refill (near-far SSSP): a drained bucket is followed by the next one INSIDE the launch
                             // while the far pile holds at most this many entries (0: every bucket change goes through the head kernel)
};

// The search is over: final counters and the elapsed device time go to the host-pinned mailbox
// BEFORE the done flag (system-scope fence in between), so a host that polls the flag can return
// without a device round trip.  One thread; c->edges_visited / vertices_visited must be final.
__device__ __forceinline__ void publish_done(const pipe_args& a, ctrl_t* c, int level) {
  long long* mb64 = reinterpret_cast<long long*>(a.mailbox + 4);
  mb64[0] = c->edges_visited;
  mb64[1] = c->vertices_visited;
  mb64[2] = (long long)wall_clock64() - c->t_start;
  a.mailbox[1] = level;
  a.mailbox[11] = c->bin_want;
  a.mailbox[12] = 0;  // the search ended in a head kernel (the many-levels body of a level kernel, grx_mid.hpp: 1)
  __threadfence_system();
  a.mailbox[0] = 1;
}

// ---------------------------------------------------------------------------
// plan: per-level bookkeeping + chunk map.  <<<1, 1024>>>
// ---------------------------------------------------------------------------
// external_control != 0: level bookkeeping, termination and counters are done by
// another kernel (direction-optimising BFS); this one only builds the chunk map of
// a top-down level and leaves when the level runs bottom-up.
// Body of the plan step for a workgroup of BLOCK threads; s_esum[0..1] must be 0 and the
// workgroup synchronised on entry.  s_wave: BLOCK / 64 + 1 ints of LDS.
// What the single-workgroup kernels need from the control block, fetched in ONE batch of loads.
// Read field by field behind early-exit branches these were four to five dependent round trips
// at the start of every head kernel (seen in the ISA: s_load / s_waitcnt / branch, repeated) --
// a third of the kernel on a level that is otherwise a handful of round trips.
struct ctrl_head {
  int level, done, nt0, nt1, total_chunks, mode, frontier_bitmap, bu_R, bu_T;
  long long edges_visited;
  __device__ __forceinline__ int nt(int parity) const { return parity ? nt1 : nt0; }
};
__device__ __forceinline__ ctrl_head load_ctrl_head(const ctrl_t* c) {
  const int4 a0 = reinterpret_cast<const int4*>(c)[0];  // level, done, n_tiles[0..1]
  const int4 a1 = reinterpret_cast<const int4*>(c)[1];  // n_items[0..1], total_chunks, pad
  ctrl_head h;
  h.mode = c->mode;
  h.frontier_bitmap = c->frontier_bitmap;
  h.bu_R = c->bu_R;
  h.bu_T = c->bu_T;
  h.edges_visited = c->edges_visited;
  h.level = a0.x;
  h.done = a0.y;
  h.nt0 = a0.z;
  h.nt1 = a0.w;
  h.total_chunks = a1.z;
  return h;
}

// What a LEVEL kernel needs from the control block, as relaxed agent-scope atomic loads: unlike
// plain loads the compiler may not sink them behind the early-exit branch, so all of them are in
// flight at once (plain loads were split into {level, done} -> branch -> {total_chunks, mode}: two
// dependent round trips in every workgroup of every level).
struct level_head {
  int level, done, nt0, nt1, total_chunks, mode;
};
__device__ __forceinline__ level_head load_level_head(const ctrl_t* c) {
  const unsigned long long ld = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&c->level),
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long nt = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&c->n_tiles[0]),
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  level_head h;
  const int tc = __hip_atomic_load(&c->total_chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int md = __hip_atomic_load(&c->mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the values are wave-uniform: hand them to the scalar unit (uniform branches, SGPR arithmetic)
  h.total_chunks = __builtin_amdgcn_readfirstlane(tc);
  h.mode = __builtin_amdgcn_readfirstlane(md);
  h.level = __builtin_amdgcn_readfirstlane((int)(unsigned)ld);
  h.done = __builtin_amdgcn_readfirstlane((int)(ld >> 32));
  h.nt0 = __builtin_amdgcn_readfirstlane((int)(unsigned)nt);
  h.nt1 = __builtin_amdgcn_readfirstlane((int)(nt >> 32));
  return h;
}

// state a plan step works from (already known to its caller, or taken from a ctrl_head)
struct plan_in {
  int done;
  int level;  // the level being planned
  int nt;     // tiles of its frontier
  int mode;
  int R, T;   // bottom-up tile ranges (external_control == 1), R == 0: dense tiles
  // binned top-down levels (grx_bin.hpp; external_control == 0 only): a level whose frontier has at
  // least bin_min out-edges runs as mode 2; the head zeroes the nb fill counters (bin_pad ints apart)
  long long bin_min = 0;
  int bin_max_degree = 0;        // ... and at most this many out-edges per frontier vertex on average (0: no limit)
  // many mid-size levels in one launch (grx_mid.hpp; external_control == 0 only): a level with at most mid_v
  // frontier vertices and mid_e out-edges runs as mode 3 (0: off)
  int mid_v = 0, mid_e = 0;
  // ... and no tile (256 frontier slots = the block ONE workgroup of that body stages) with more than mid_tile_e out-edges
  // (0: no such rule).  Round 5, last session: a search from a low-degree vertex of a scale-free graph meets, two or three
  // levels in, a frontier of ~10 vertices with 10-30 k out-edges -- it qualified (<= mid_v vertices, <= mid_e edges) and ONE
  // workgroup then walked its 11-16 chunks one after the other, ~12 us each, where the regular level kernel spreads them over as
  // many workgroups: the three sources of bench.py's multi_source section with such a level were its slowest in every round
  // (85-97 GTEPS against 125-150).  The body is written for road-like frontiers: a tile of 256 vertices there has < 1 k edges.
  int mid_tile_e = 0;
  // EARLY levels -- less than a quarter of the graph visited so far -- are binned from bin_min / bin_early_div out-edges on
  // (1: same threshold everywhere).  The threshold went from 2^20 to 2^21 this round because a LATE level of the deep stand-in
  // (1.29 M edges, next to nothing left to discover) is cheaper on the claim-per-edge advance; an early level of that size
  // discovers a third of its targets, and each discovery costs the claim-per-edge body a successful atomic, a compaction slot
  // and its share of a tile reservation (source 269895 of multi_source: level 918 vertices / 1.71 M edges -> 563 k discoveries,
  // 140 GTEPS with the level binned in round 4, 108-117 with the one threshold of round 5).
  int bin_early_div = 1;
  // the launch group of this level carries the scatter / sweep kernels (the host leaves them out of groups where the
  // previous search on the graph had no fat level); seq: index of the group
  int bin_allowed = 1, seq = 0;
  int bin_forced = 0;            // the group has no level kernel (exact schedule, grx_graph::bin_exact): mode 2 whatever the size
  int only_finish = 0;           // the group has NO kernel behind its head (the last group of a repeated search): a search that is
                                 // not over is left exactly as it is -- the head of the next group plans the level
  int32_t* bin_fill = nullptr;
  int32_t* bin_queue = nullptr;  // per-XCD claim queue heads (16 slots, bin_pad apart), zeroed with the fill counters
  int bin_nb = 0, bin_pad = 0;
  // partitioned searches (round 6, grx_bfs_kernels.hpp part_args): the search is over when the frontier of ALL ranks is empty
  // (g_n = its all-reduced size; -1: single GPU, this rank's tile count decides) -- a rank whose own share is empty still
  // counts the level -- and "a quarter of the graph" means a quarter of this rank's share
  long long g_n = -1;
  int part_P = 1;
};

template <int BLOCK>
__device__ __forceinline__ void plan_body(const pipe_args& a, ctrl_t* c, int external_control, int* s_wave,
                                          unsigned long long* s_esum, const plan_in& in) {
  const int tid = threadIdx.x;
  const int done = in.done;
  const int level = in.level;
  const int p = level & 1;
  const int nt = in.nt;
  const int mode = in.mode;
  if (done) return;
  if (external_control == 1 && mode != 0) return;
  if (!external_control && (in.g_n >= 0 ? in.g_n == 0 : nt == 0)) {
    if (tid == 0) {
      c->done = 1;
      c->level = level;  // == number of advance iterations executed
      publish_done(a, c, level);
    }
    return;
  }
  if (in.only_finish) return;
  if (in.bin_min > 0) {
    if (tid < in.bin_nb) in.bin_fill[tid * in.bin_pad] = 0;
    if (tid < 16) in.bin_queue[tid * in.bin_pad] = 0;
  }
  long long esum = 0, vsum = 0;  // traversed edges / frontier vertices of the level (two 64-bit sums)
  int mine = 0, carry;
  const int R = external_control == 1 ? in.R : 0;
  if (R > 0) {
    // The frontier was left by a bottom-up level: workgroup r of that launch filled the first
    // n_r indices of its static tile range [r * T, (r + 1) * T) (n_r: bu_part word 0 >> 40).
    // Walk the ranges, not the ~20 k mostly empty tile indices.
    const int T = in.T;
    // thread t owns the consecutive ranges [t * RPT, (t + 1) * RPT), RPT = ceil(R / BLOCK): any
    // number of bottom-up workgroups is covered (the chunk map stays in range order)
    const int RPT = (R + BLOCK - 1) / BLOCK;
    const int r0 = tid * RPT, r1 = min(R, r0 + RPT);
    for (int r = r0; r < r1; ++r) {
      const int nr = (int)(a.bu_part[4 * r] >> 40);
      const int base = r * T;
      for (int k = 0; k < nr; ++k) mine += a.tile_chunks[base + k];
    }
    int pre = dev::block_exclusive_sum<BLOCK>(mine, s_wave, &carry);
    int2* map = reinterpret_cast<int2*>(a.chunk_tile);
    for (int r = r0; r < r1; ++r) {
      const int nr = (int)(a.bu_part[4 * r] >> 40);
      const int base = r * T;
      for (int k = 0; k < nr; ++k) {
        const int ch = a.tile_chunks[base + k];
        for (int j = 0; j < ch; ++j) map[pre + j] = make_int2(base + k, j);
        pre += ch;
      }
    }
  } else if (!external_control && in.bin_min > 0 && c->map_level == level) {
    // The producers of this frontier (sweep claim of a binned level, grx_bin.hpp) wrote the chunk map and summed the
    // counters themselves: nothing to walk.  (Planning 8 k tiles / 18 k chunks here took 33 us on the LJ stand-in,
    // 72 k chunks 63 us on the kron stand-in.)
    carry = c->map_chunks;
    if (tid == 0) {
      esum = c->q_edges[p];
      vsum = c->n_items[p];
    }
  } else {
  // Thread t owns the contiguous tiles [t * K, (t + 1) * K), K = ceil(nt / BLOCK): all its loads
  // are independent (issued G at a time), ONE block scan serves the whole level, and a second
  // sweep over the same (now cached) counts writes the chunk map.
  const int K = (nt + BLOCK - 1) / BLOCK;
  const int t0 = tid * K, t1 = min(nt, t0 + K);
  constexpr int G = 8;
  for (int i0 = t0; i0 < t1; i0 += G) {
    int ch[G], ts[G], tc[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const int i = min(i0 + k, t1 - 1);
      ch[k] = a.tile_chunks[i];
      ts[k] = a.tile_sums[i];
      tc[k] = a.tile_count[i];
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
      if (i0 + k < t1) {
        mine += ch[k];
        esum += (long long)ts[k];
        vsum += (long long)tc[k];
        if (in.mid_tile_e > 0 && ts[k] > in.mid_tile_e) vsum += 1ll << 32;  // tiles too heavy for the many-levels body: counted
                                                                             // above the vertex count (< 2^31) of the same sum
      }
    }
  }
  int pre = dev::block_exclusive_sum<BLOCK>(mine, s_wave, &carry);
  {
    // chunk -> {tile, chunk index inside the tile}: ONE 8-byte load per chunk in the level kernel.
    // The entries of a tile are written by the 64 lanes of its owner's WAVE together: a tile of hubs has
    // hundreds to thousands of chunks (the frontier right behind a hub source), and one lane storing them one
    // after the other made this kernel 30-60 us on the levels where it matters.
    int2* map = reinterpret_cast<int2*>(a.chunk_tile);
    const int lane = dev::lane_id();
    for (int k = 0; k < K; ++k) {  // K is uniform; lanes past t1 contribute empty tiles
      const int i = t0 + k;
      const int ch = i < t1 ? a.tile_chunks[i] : 0;
      const int any_big = dev::ballot(ch > 4) != 0ull;
      if (!any_big) {
        for (int j = 0; j < ch; ++j) map[pre + j] = make_int2(i, j);
      } else {
        for (int l = 0; l < 64; ++l) {
          const int c_l = __shfl(ch, l, 64);
          if (c_l == 0) continue;  // uniform
          const int i_l = __shfl(i, l, 64);
          const int p_l = __shfl(pre, l, 64);
          for (int j = lane; j < c_l; j += 64) map[p_l + j] = make_int2(i_l, j);
        }
      }
      pre += ch;
    }
  }
  }
  // 64-bit block reductions of the traversed-edge and frontier-vertex counts (s_esum[0], [1])
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    esum += __shfl_xor(esum, o, 64);
    vsum += __shfl_xor(vsum, o, 64);
  }
  if (dev::lane_id() == 0) {
    atomicAdd(&s_esum[0], (unsigned long long)esum);
    atomicAdd(&s_esum[1], (unsigned long long)vsum);
  }
  __syncthreads();
  if (tid == 0) {
    c->total_chunks = carry;
    if (external_control != 1) {
      const long long edges = (long long)s_esum[0];
      const int nitems = (int)(s_esum[1] & 0xffffffffull);
      const int heavy_tiles = (int)(s_esum[1] >> 32);  // (plan_in::mid_tile_e)
      c->edges_visited += edges;
      c->vertices_visited += nitems;
      c->n_items[p] = nitems;
      c->q_edges[p] = edges;
      if (in.bin_min > 0 || in.mid_v > 0) {
        int mode = 0;
        const bool early = in.bin_early_div > 1 && c->vertices_visited * 4 * in.part_P < (long long)a.V;  // (this frontier included)
        const long long bin_from = early ? in.bin_min / in.bin_early_div : in.bin_min;
        const bool fat = in.bin_min > 0 && edges >= bin_from &&
                         (in.bin_max_degree <= 0 || edges <= (long long)in.bin_max_degree * nitems);
        if (fat && in.seq < 32) c->bin_want |= 1 << in.seq;
        if ((fat || in.bin_forced) && in.bin_allowed)
          mode = 2;
        // (the tile queue must be reasonably dense too: the few workgroups of that body walk it themselves, and a
        // level of the regular kernels on a wide grid leaves thousands of nearly empty tiles behind -- such a level
        // is expanded by the regular kernels once more, which compacts it)
        else if (!fat && !in.bin_forced && in.mid_v > 0 && nitems > 0 && nitems <= in.mid_v && edges <= (long long)in.mid_e &&
                 heavy_tiles == 0 && nt <= 4 * ((nitems + TILE - 1) / TILE) + 256)
          mode = 3;
        c->mode = mode;
        if (mode == 2) {  // the sweep claim of this level accumulates the next level's counters and chunk map
          c->map_chunks = 0;
          c->n_items[p ^ 1] = 0;
          c->q_edges[p ^ 1] = 0;
        }
        if (mode == 3) {
          c->mid_bar = 0;
          c->mid_reg = 0u;
          c->mid_G = 0;
          c->mid_cnt[0] = 0;
          c->mid_cnt[1] = 0;
          c->mid_cnt[2] = 0;
          if (a.mid_flags)
            for (int i = 0; i < MID_FLAG_WORDS; ++i) a.mid_flags[i] = 0ull;
        }
      }
      if (!external_control) {
        c->level = level;
        c->n_tiles[p ^ 1] = 0;
        a.mailbox[1] = level;
        a.mailbox[2] = nitems;
      }
    }
  }
}

template <int BLOCK>
__device__ __forceinline__ void plan_body(const pipe_args& a, ctrl_t* c, int external_control, int* s_wave,
                                          unsigned long long* s_esum) {
  const ctrl_head h = load_ctrl_head(c);
  plan_in in;
  in.done = h.done;
  in.level = external_control ? h.level : h.level + 1;
  in.nt = h.nt(in.level & 1);
  in.mode = h.mode;
  in.R = h.bu_R;
  in.T = h.bu_T;
  plan_body<BLOCK>(a, c, external_control, s_wave, s_esum, in);
}

static __global__ __launch_bounds__(PLAN_BLOCK) void plan_kernel(pipe_args a, int external_control) {
  __shared__ int s_wave[PLAN_BLOCK / 64 + 1];
  __shared__ unsigned long long s_esum[2];
  if (threadIdx.x < 2) s_esum[threadIdx.x] = 0ull;
  __syncthreads();
  plan_body<PLAN_BLOCK>(a, a.ctrl, external_control, s_wave, s_esum);
}

// Emit n (<= TILE) vertices s_out[lo .. lo+n) as one tile of the frontier with
// parity q; its degree sum is computed here so the next level needs no scan over
// the frontier.  Tile indices are RESERVED TILE_RESERVE at a time: one global
// atomic per 4 x 256 emitted vertices (all workgroups hit the same counter word,
// and a single word sustains only ~90 atomics/us on this part).  Per-tile
// vertex counts and degree sums are summed by the next plan/decide kernel
// instead of being accumulated with more atomics.
// Block-wide call (contains __syncthreads); s_wave / s_res are LDS scratch,
// s_res = {next reserved tile, end of reservation, tile of this emit}.
__device__ __forceinline__ void emit_tile(const pipe_args& a, ctrl_t* c, int q, const int* s_out, int lo, int n,
                                          int* s_wave, int* s_res) {
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  const int wid = tid >> 6;
  // the reservation atomic is issued FIRST so that it travels together with the degree loads
  // (one round trip instead of two on the tail of every small level)
  int fresh = -1;
  if (tid == 0 && s_res[0] == s_res[1]) fresh = atomicAdd(&c->n_tiles[q], TILE_RESERVE);
  int x = -1, deg = 0;
  if (tid < n) {
    x = s_out[lo + tid];
    deg = a.ro[x + 1] - a.ro[x];
  }
  const int tsum = dev::wave_sum(deg);
  if (lane == 0) s_wave[wid] = tsum;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
#pragma unroll
    for (int i = 0; i < ADV_BLOCK / 64; ++i) tot += s_wave[i];
    if (fresh >= 0) {
      s_res[0] = fresh;
      s_res[1] = fresh + TILE_RESERVE;
    }
    const int tix = s_res[0]++;
    a.tile_sums[tix] = tot;
    a.tile_chunks[tix] = (tot + CHUNK - 1) / CHUNK;
    a.tile_count[tix] = n;
    s_res[2] = tix;
  }
  __syncthreads();
  a.frontier[q][(size_t)s_res[2] * TILE + tid] = x;
}

// k (<= MAX_EMIT) FULL tiles s_out[lo .. lo + k * TILE) at once: all degree loads of the k
// tiles are issued together, the k tile indices come from (at most) one global atomic, and
// the block synchronises twice per CALL -- emit_tile costs two dependent round trips and
// three barriers per TILE, which made a chunk that discovers 2000 vertices (the first level
// from a hub) spend most of its time emitting.  Block-wide call.
constexpr int MAX_EMIT = (CHUNK + TILE - 1) / TILE + 1;
struct emit_smem {
  int sum[MAX_EMIT][ADV_BLOCK / 64];
  int tix[MAX_EMIT];
};
__device__ __forceinline__ void emit_full_tiles(const pipe_args& a, ctrl_t* c, int q, const int* s_out, int lo, int k,
                                                emit_smem& es, int* s_res) {
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  const int wid = tid >> 6;
  int x[MAX_EMIT], deg[MAX_EMIT];
  // tile indices: what is left of the current reservation first, then ONE new reservation, whose
  // atomic is issued before the degree loads so that both travel together
  int fresh = -1, take = 0;
  if (tid == 0) {
    const int have = s_res[1] - s_res[0];
    if (have < k) {
      take = (k - have + TILE_RESERVE - 1) / TILE_RESERVE * TILE_RESERVE;
      fresh = atomicAdd(&c->n_tiles[q], take);
    }
  }
#pragma unroll
  for (int j = 0; j < MAX_EMIT; ++j) x[j] = s_out[lo + (j < k ? j : 0) * TILE + tid];
#pragma unroll
  for (int j = 0; j < MAX_EMIT; ++j) deg[j] = a.ro[x[j] + 1] - a.ro[x[j]];  // unconditional: x[j] is a valid vertex
#pragma unroll
  for (int j = 0; j < MAX_EMIT; ++j) {
    const int t = dev::wave_sum(j < k ? deg[j] : 0);
    if (lane == 0) es.sum[j][wid] = t;
  }
  if (tid == 0) {
    int have = s_res[1] - s_res[0];
    int j = 0;
    for (; j < k && have > 0; ++j, --have) es.tix[j] = s_res[0]++;
    if (j < k) {
      int base = fresh;
      s_res[1] = base + take;
      for (; j < k; ++j) es.tix[j] = base++;
      s_res[0] = base;
    }
  }
  __syncthreads();
  if (tid < k) {
    int tot = 0;
#pragma unroll
    for (int i = 0; i < ADV_BLOCK / 64; ++i) tot += es.sum[tid][i];
    const int tix = es.tix[tid];
    a.tile_sums[tix] = tot;
    a.tile_chunks[tix] = (tot + CHUNK - 1) / CHUNK;
    a.tile_count[tix] = TILE;
  }
#pragma unroll
  for (int j = 0; j < MAX_EMIT; ++j)
    if (j < k) a.frontier[q][(size_t)es.tix[j] * TILE + tid] = x[j];
  __syncthreads();
}

// Reserved but unused tile indices become empty tiles (never staged: 0 chunks).
__device__ __forceinline__ void release_tiles(const pipe_args& a, const int* s_res) {
  const int t = s_res[0] + (int)threadIdx.x;
  if (t < s_res[1]) {
    a.tile_sums[t] = 0;
    a.tile_chunks[t] = 0;
    a.tile_count[t] = 0;
  }
}

// ---------------------------------------------------------------------------
// advance + fused compaction.  Policy interface (all __device__):
//   void begin(ctrl_t*)                               once per workgroup
//   src_state load_source(int v)                      per staged slot (e.g. dist[v])
//   bool precheck(src_state, int nbr, int e, int& cand)
//        cheap READ-ONLY filter; called for every lane with a valid (nbr, e) -- lanes past
//        the end of the chunk pass edge 0 -- so its loads are unconditional; `cand` carries a
//        per-edge value (e.g. the tentative distance bits) to the later phases
//   int  claim(int nbr, int cand)                     the claiming atomic; returns its raw result
//   [two_claims] bool need2(int raw1, int cand), int claim2(int nbr)   a second, dependent atomic
//   int  code(int raw1, int raw2, int nbr, int cand)  1 => nbr joins the output,
//                                                     2 => nbr goes to the policy's SIDE pile
//                                                     (policies with `has_side`), 0 => dropped
//   int  visit(src_state, int nbr, int e)             precheck-survivor -> code, the whole chain for
//                                                     one edge (tiny_levels_body)
//   side pile (has_side): side_reserve(ctrl, n) -> base index or -1, side_store(i, v) -> key,
//                         side_commit(min key of a wave)
//
// WHY PHASES.  A memory operation under a per-lane condition lives in its own basic block, and
// when its result is consumed in that block the compiler emits s_waitcnt vmcnt(0) right behind
// it: the ADV_ITEMS probes of a lane, and then its ADV_ITEMS atomics, became 16 SERIAL round
// trips per chunk (seen in the ISA of the first version of this kernel).  Here every phase
// issues the operations of all ADV_ITEMS edges -- pure loads unconditionally from clamped
// indices, atomics conditionally but with their raw result consumed only by the NEXT phase --
// so a chunk costs one round trip per phase.
// ---------------------------------------------------------------------------
template <class Policy, class = void>
struct policy_has_side : std::false_type {};
template <class Policy>
struct policy_has_side<Policy, std::void_t<decltype(Policy::has_side)>> : std::bool_constant<Policy::has_side> {};

// optional hooks: on_accept(nbr) for every vertex that joins the output; tiny_enter(list, n) /
// tiny_hand_back(level, lds list, n, cap, spill) around tiny_levels_body
template <class Policy, class = void>
struct policy_has_accept : std::false_type {};
template <class Policy>
struct policy_has_accept<Policy, std::void_t<decltype(&Policy::on_accept)>> : std::true_type {};
template <class Policy, class = void>
struct policy_two_claims : std::false_type {};
template <class Policy>
struct policy_two_claims<Policy, std::void_t<decltype(Policy::two_claims)>> : std::bool_constant<Policy::two_claims> {};

// LDS of one advance workgroup.
template <class Policy>
struct advance_smem {
  static constexpr bool SIDE = policy_has_side<Policy>::value;
  int side[SIDE ? (CHUNK + ADV_BLOCK) : 1];
  int side_cnt;
  int side_base;
  int seg[TILE + 1];
  int start[TILE];
  int src[TILE];
  typename Policy::src_state state[TILE];
  int out[TILE + CHUNK];
  int wave[ADV_BLOCK / 64 + 1];
  int cnt;
  int res[3];
  emit_smem emit;
};

// Store the sc side-pile entries of a workgroup at [sb, sb + sc) and commit the minimum of the
// keys the policy returns with one atomic per wave.
template <class Policy>
__device__ __forceinline__ void side_flush(Policy& pol, const int* side, int sb, int sc) {
  unsigned key = 0xffffffffu;
  for (int i = threadIdx.x; i < sc; i += ADV_BLOCK) key = min(key, pol.side_store(sb + i, side[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o, 64));
  if (dev::lane_id() == 0 && key != 0xffffffffu) pol.side_commit(key);
}

// The work of one advance workgroup on one level: chunks chunk_first, chunk_first +
// chunk_stride, ... of the frontier with parity p; winners are emitted as tiles of parity
// p ^ 1.  FRESH: re-read the frontier slots past the CU's L1 (for callers whose slots were
// written by the same workgroup earlier in the same launch).
template <class Policy, bool FRESH>
__device__ __forceinline__ void advance_block(const pipe_args& a, ctrl_t* c, Policy& pol, advance_smem<Policy>& sm,
                                              int p, int chunk_first, int chunk_stride, int total_chunks,
                                              const int* chunk_tile) {
  constexpr bool SIDE = policy_has_side<Policy>::value;
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  const int32_t* in = a.frontier[p];
  if (tid == 0) { sm.cnt = 0; sm.res[0] = 0; sm.res[1] = 0; sm.side_cnt = 0; }
  __syncthreads();

  // total_chunks < 0: TILE MODE -- no chunk map was built for this level (the level after a
  // bottom-up -> top-down switch: its queue is rebuilt by many workgroups, and a last-
  // workgroup plan would need agent-scope fences, i.e. L2 write-backs on a multi-XCD part).
  // Workgroup w takes tiles w, w + stride, ... and walks each tile's chunks itself.
  const bool tile_mode = total_chunks < 0;
  const int n_units = tile_mode ? c->n_tiles[p] : total_chunks;
  for (int unit = chunk_first; unit < n_units; unit += chunk_stride)
  for (int lc_t = 0, n_lc = tile_mode ? a.tile_chunks[unit] : 1; lc_t < n_lc; ++lc_t) {
    int t = unit, lc = lc_t;
    if (!tile_mode) {
      const int2 tl = reinterpret_cast<const int2*>(chunk_tile)[unit];
      t = tl.x;
      lc = tl.y;
    }
    // ---- stage the tile -------------------------------------------------
    int v;
    if constexpr (FRESH)
      v = __hip_atomic_load(&in[(size_t)t * TILE + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      v = in[(size_t)t * TILE + tid];
    int rs = 0, deg = 0;
    typename Policy::src_state st{};
    if (v >= 0) {
      rs = a.ro[v];
      deg = a.ro[v + 1] - rs;
      st = pol.load_source(v);
    }
    int tot;
    const int ex = dev::block_exclusive_sum<ADV_BLOCK>(deg, sm.wave, &tot);
    sm.seg[tid] = ex;
    sm.start[tid] = rs;
    sm.src[tid] = v;
    sm.state[tid] = st;
    if (tid == 0) sm.seg[TILE] = tot;
    __syncthreads();

    // ---- atoms of this chunk ---------------------------------------------
    const int a0 = lc * CHUNK;
    const int a_end = min(tot, a0 + CHUNK);
    int e_k[ADV_ITEMS], slot_k[ADV_ITEMS], n_k[ADV_ITEMS], cand_k[ADV_ITEMS];
#pragma unroll
    for (int k = 0; k < ADV_ITEMS; ++k) {
      const int atom = a0 + k * ADV_BLOCK + tid;
      int lo = 0;
      if (atom < a_end) {
#pragma unroll
        for (int step = TILE / 2; step >= 1; step >>= 1)
          if (sm.seg[lo + step] <= atom) lo += step;
        e_k[k] = sm.start[lo] + (atom - sm.seg[lo]);
      } else {
        e_k[k] = -1;
      }
      slot_k[k] = lo;
    }
    // phase 0: column indices.  Lanes past the end of the chunk read edge 0 (the chunk has at
    // least one atom, so the graph has an edge), which keeps every later pure load valid.
#pragma unroll
    for (int k = 0; k < ADV_ITEMS; ++k) n_k[k] = a.ci[e_k[k] >= 0 ? e_k[k] : 0];
    // phase 1: read-only filter
    bool pre_k[ADV_ITEMS];
#pragma unroll
    for (int k = 0; k < ADV_ITEMS; ++k) {
      const bool ok = e_k[k] >= 0;
      cand_k[k] = 0;
      const bool pass = pol.precheck(sm.state[slot_k[k]], n_k[k], ok ? e_k[k] : 0, cand_k[k]);
      pre_k[k] = pass & ok;
    }
    // phase 2: the claiming atomics, all issued before any result is looked at
    int r1_k[ADV_ITEMS], r2_k[ADV_ITEMS];
#pragma unroll
    for (int k = 0; k < ADV_ITEMS; ++k) {
      r1_k[k] = 0;
      r2_k[k] = 0;
      if (pre_k[k]) r1_k[k] = pol.claim(n_k[k], cand_k[k]);
    }
    if constexpr (policy_two_claims<Policy>::value) {
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        const bool need = pre_k[k] & pol.need2(r1_k[k], cand_k[k]);
        if (need) r2_k[k] = pol.claim2(n_k[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < ADV_ITEMS; ++k) {
      int code = 0;
      if (pre_k[k]) code = pol.code(r1_k[k], r2_k[k], n_k[k], cand_k[k]);
      const bool keep = code == 1;
      const unsigned long long m = dev::ballot(keep);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&sm.cnt, __popcll(m));
        base = dev::wave_bcast0(base);
        if (keep) {
          sm.out[base + dev::mask_rank(m)] = n_k[k];
          if constexpr (policy_has_accept<Policy>::value) pol.on_accept(n_k[k]);
        }
      }
      if constexpr (SIDE) {
        const bool aside = code == 2;
        const unsigned long long ms = dev::ballot(aside);
        if (ms) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&sm.side_cnt, __popcll(ms));
          base = dev::wave_bcast0(base);
          if (aside) sm.side[base + dev::mask_rank(ms)] = n_k[k];
        }
      }
    }
    __syncthreads();
    if constexpr (SIDE) {
      // flush the side pile when it could overflow on the next chunk
      const int sc = sm.side_cnt;
      if (sc >= ADV_BLOCK) {
        if (tid == 0) sm.side_base = pol.side_reserve(c, sc);
        __syncthreads();
        const int sb = sm.side_base;
        if (sb >= 0) side_flush(pol, sm.side, sb, sc);
        __syncthreads();
        if (tid == 0) sm.side_cnt = 0;
        __syncthreads();
      }
    }
    // ---- flush full tiles --------------------------------------------------
    int cnt = sm.cnt;
    if (cnt >= TILE) {
      // the LAST k * TILE entries leave; the first cnt % TILE stay for the next chunk
      const int k = cnt / TILE;
      emit_full_tiles(a, c, p ^ 1, sm.out, cnt - k * TILE, k, sm.emit, sm.res);
      cnt -= k * TILE;
    }
    if (tid == 0) sm.cnt = cnt;
    __syncthreads();
  }
  const int rem = sm.cnt;
  if (rem > 0) emit_tile(a, c, p ^ 1, sm.out, 0, rem, sm.wave, sm.res);
  __syncthreads();
  release_tiles(a, sm.res);
  if constexpr (SIDE) {
    const int sc = sm.side_cnt;
    if (sc > 0) {
      if (tid == 0) sm.side_base = pol.side_reserve(c, sc);
      __syncthreads();
      const int sb = sm.side_base;
      if (sb >= 0) side_flush(pol, sm.side, sb, sc);
    }
  }
  __syncthreads();
}

template <class Policy>
__global__ __launch_bounds__(ADV_BLOCK) void advance_kernel(pipe_args a, Policy pol) {
  __shared__ advance_smem<Policy> sm;
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);
  if (h.done) return;
  if (h.mode != 0) return;  // this level runs bottom-up
  const int p = h.level & 1;
  pol.begin(c);
  advance_block<Policy, false>(a, c, pol, sm, p, blockIdx.x, gridDim.x, h.total_chunks, a.chunk_tile);
}

// ---------------------------------------------------------------------------
// Many TINY levels in one launch.  A single 1024-thread workgroup keeps the frontier in
// LDS (<= 4096 vertices and <= 4096 out-edges per level) and plays plan + advance itself, so
// a level costs a handful of dependent global round trips (row offsets -> column index ->
// label probe -> claim) instead of a group of kernel launches (12.6 us per level measured for
// head + level kernel).  It absorbs the first and last levels of scale-free searches and whole
// chain-like searches.  It starts from the tiled queue the regular kernels left, and when a
// level is too big (or could switch direction) it hands the frontier back as tiles and leaves
// the control block exactly as the plan / decide step of the head kernels expects it.  <<<1, 1024>>>
// The edges of a level are processed with the same phase structure as advance_block (all
// loads of a phase in flight together).
// SIZING (measured, road stand-in, 4.6 k vertices / 11 k edges per level on average): raising
// the caps to 8192 vertices / 16384 edges made the search 2x SLOWER (26 us per level): one CU
// retires about one scattered memory transaction every two clocks, so ~30 k transactions per
// level belong on 256 CUs, not on one -- the single workgroup only wins while a level is a few
// thousand transactions.
// ---------------------------------------------------------------------------
constexpr int TINY_THREADS = 1024;
constexpr int TINY_EDGES = 4096;       // out-edges per level the workgroup is willing to take
constexpr int TINY_MAX_TILES = 512;    // tiles the entry gather is willing to look at (a decline past
                                       // this check costs the level a round trip over the tile counts)

template <class Policy, class = void>
struct policy_stateless : std::false_type {};
template <class Policy>
struct policy_stateless<Policy, std::void_t<decltype(Policy::stateless)>> : std::bool_constant<Policy::stateless> {};

template <class Policy>
struct tiny_smem {
  static constexpr bool STATELESS = policy_stateless<Policy>::value;
  // frontier vertices resident in LDS.  3072, not 4096: the head kernels must stay BELOW 64 KB of LDS -- a kernel that
  // asks for more pays 10-20 us on EVERY launch on this part (measured on sssp_nf_level_kernel: shortest launch 2.8 us
  // with 51 KB, 13-25 us with 82 KB), and these kernels are launched once per level
  static constexpr int CAP = 3072;
  static constexpr int ITEMS = 4;    // atoms per thread per pass: one pass covers TINY_EDGES
  static_assert(CAP + 1 >= TINY_MAX_TILES + 1, "seg[] doubles as the tile-offset scratch of the entry gather");
  int buf[2][CAP];
  int seg[CAP + 1];
  int start[CAP];
  typename Policy::src_state state[STATELESS ? 1 : CAP];
  int wave[TINY_THREADS / 64 + 1];
  int n;
  unsigned long long total;
};

// Hand a frontier of n vertices back to the regular path as tiles of parity p: entries below
// CAP come from LDS (`lds`), the others already sit in a.frontier[p] at their final slots
// (written earlier in this launch by this workgroup: re-read past the L1).  Degrees are read
// here.  Leaves c->level = level - 1 so that the next plan / decide step runs `level`.
template <class Policy>
__device__ __forceinline__ void tiny_hand_back(const pipe_args& a, ctrl_t* c, const int* lds, int n, int level,
                                               long long edges_total, long long vertices_total,
                                               tiny_smem<Policy>& sm) {
  constexpr int CAP = tiny_smem<Policy>::CAP;
  const int tid = threadIdx.x;
  const int p = level & 1;
  const int tiles = (n + TILE - 1) / TILE;
  int32_t* out = a.frontier[p];
  for (int base = 0; base < tiles * TILE; base += TINY_THREADS) {
    const int slot = base + tid;
    int v = -1;
    if (slot < n) v = slot < CAP ? lds[slot] : __hip_atomic_load(&out[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int vv = v >= 0 ? v : lds[0];
    int deg = a.ro[vv + 1] - a.ro[vv];
    if (v < 0) deg = 0;
    if (slot < tiles * TILE && (slot < CAP || slot >= n)) out[slot] = v;
    const int wsum = dev::wave_sum(deg);
    if (dev::lane_id() == 0) sm.wave[tid >> 6] = wsum;
    __syncthreads();
    if ((tid & (TILE - 1)) == 0 && slot < tiles * TILE) {
      const int w0 = tid >> 6;
      const int sum = sm.wave[w0] + sm.wave[w0 + 1] + sm.wave[w0 + 2] + sm.wave[w0 + 3];
      const int t = slot / TILE;
      a.tile_sums[t] = sum;
      a.tile_chunks[t] = (sum + CHUNK - 1) / CHUNK;
      a.tile_count[t] = min(TILE, n - t * TILE);
    }
    __syncthreads();
  }
  if (tid == 0) {
    c->n_tiles[p] = tiles;
    c->level = level - 1;
    c->edges_visited = edges_total;
    c->vertices_visited += vertices_total;
  }
  __syncthreads();  // a plan/decide step may follow in the same workgroup
}

// Returns 1 when the search finished inside (done is set); 0 when it declined without touching
// anything; 2 when it ran levels and handed the frontier back as tiles (control block changed):
// for 0 and 2 the regular per-level path has to continue.
template <class Policy>
__device__ __forceinline__ int tiny_levels_body(const pipe_args& a, Policy& pol, int do_enabled,
                                                long long n_edges_total, tiny_smem<Policy>& sm, const ctrl_head& h) {
  constexpr int CAP = tiny_smem<Policy>::CAP;
  constexpr bool STATELESS = tiny_smem<Policy>::STATELESS;
  constexpr int TINY_ITEMS = tiny_smem<Policy>::ITEMS;
  ctrl_t* c = a.ctrl;
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  if (h.done) return 1;
  if (h.frontier_bitmap) return 0;  // partitioned BFS: the frontier is a bitmap right now
  // the previous level ran bottom-up: its discoveries sit in static per-workgroup tile ranges and
  // its counters in bu_part records, which only the decide / plan steps know how to read (and the
  // hand-back paths below leave mode / bu_R untouched).  On a 256-CU part the ranges span more
  // than TINY_MAX_TILES indices anyway; on a smaller grid they might not.  (mode 2, a binned
  // top-down level, leaves ordinary dense tiles.)
  // (Round 5: `mode` alone decides.  bu_R stays set through the FIRST top-down level behind a bottom-up one -- its plan step
  // reads the ranges -- but what that level emits are ordinary tiles: declining on bu_R > 0 cost a direction-optimising
  // search on the LJ stand-in a launch pair for each of its last two levels, 49 and 1 vertices.  Every reader of bu_R is
  // guarded by mode: bfs_head_kernel `in.R = h.mode ? h.bu_R : 0`, bfs_decide_body zeroes it behind a top-down level.)
  if (h.mode == 1) return 0;
  int level = h.level + 1;          // next level to run
  {
    // ---- entry: gather the tiled queue into LDS ---------------------------------------
    const int p = level & 1;
    const int nt = h.nt(p);
    // sparse tiles are common (every producing workgroup leaves a partial tile and reserves
    // tile ids four at a time): go by the VERTEX count, over a bounded number of tiles
    if (nt > TINY_MAX_TILES) return 0;
    if (tid == 0) sm.total = 0ull;
    __syncthreads();
    constexpr int TPT = (TINY_MAX_TILES + TINY_THREADS - 1) / TINY_THREADS;  // tiles per thread, consecutive
    int cnt_k[TPT], mine = 0;
    {
      // vertices in the low 32 bits, out-edges (known per tile) in the high 32
      unsigned long long cnt = 0;
#pragma unroll
      for (int k = 0; k < TPT; ++k) {
        const int t = tid * TPT + k;
        cnt_k[k] = t < nt ? a.tile_count[t] : 0;
        const int es = t < nt ? a.tile_sums[t] : 0;
        mine += cnt_k[k];
        cnt += (unsigned long long)cnt_k[k] + ((unsigned long long)es << 32);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
      if (lane == 0 && cnt) atomicAdd(&sm.total, cnt);
    }
    __syncthreads();
    const unsigned long long total = sm.total;
    __syncthreads();
    if ((int)(total & 0xffffffffull) > CAP || (long long)(total >> 32) > TINY_EDGES) return 0;
    // tiles are front-packed (tile_count valid slots, then -1): exclusive scan of the counts
    // gives every tile its place in LDS, then one wave copies one tile
    int tot;
    int ex = dev::block_exclusive_sum<TINY_THREADS>(mine, sm.wave, &tot);
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int t = tid * TPT + k;
      if (t < nt) sm.seg[t] = ex;
      ex += cnt_k[k];
    }
    __syncthreads();
    const int32_t* in = a.frontier[p];
    for (int t = tid >> 6; t < nt; t += TINY_THREADS / 64) {
      const int cnt = a.tile_count[t], off = sm.seg[t];
#pragma unroll
      for (int j = 0; j < TILE / 64; ++j) {
        const int s_ = j * 64 + lane;
        if (s_ < cnt) sm.buf[0][off + s_] = in[(size_t)t * TILE + s_];
      }
    }
    if (tid == 0) sm.n = tot;
    __syncthreads();
  }
  int n = sm.n;
  int sel = 0;
  if constexpr (policy_has_accept<Policy>::value) pol.tiny_enter(sm.buf[0], n);
  long long edges_done = 0, vertices_done = 0;
  const long long edges_before = h.edges_visited;
  constexpr int PER = CAP / TINY_THREADS;  // frontier slots per thread in the degree scan
  static_assert(CAP % TINY_THREADS == 0, "whole slots per thread");
  constexpr int SEARCH0 = CAP > 2048 ? 2048 : (CAP > 1024 ? 1024 : 512);  // first step of the owner search: largest power of two below CAP
  static_assert(SEARCH0 < CAP && 2 * SEARCH0 >= CAP, "the steps must reach every slot");
  for (;;) {
    const int* cur = sm.buf[sel];
    int* nxt = sm.buf[sel ^ 1];
    if (n == 0) {
      if (tid == 0) {
        c->done = 1;
        c->level = level;
        c->edges_visited = edges_before + edges_done;
        c->vertices_visited += vertices_done;
        publish_done(a, c, level);
      }
      return 1;
    }
    if (n > CAP) {
      // the level just finished discovered more than LDS holds (the overflow went straight to
      // the tile array): back to the regular kernels
      if constexpr (policy_has_accept<Policy>::value) pol.tiny_hand_back(level, cur, n, CAP, a.frontier[level & 1]);
      tiny_hand_back<Policy>(a, c, cur, n, level, edges_before + edges_done, vertices_done, sm);
      return 2;
    }
    // ---- degrees + exclusive scan (each thread owns PER consecutive slots) ----------
    // all row-offset loads of the thread in flight together: slots past n read vertex cur[0]
    int rs[PER], deg[PER], local = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid * PER + k;
      const int v = cur[i < n ? i : 0];
      rs[k] = a.ro[v];
      deg[k] = a.ro[v + 1];
      if constexpr (!STATELESS) {
        const auto st = pol.load_source(v);
        if (i < n) sm.state[i] = st;
      }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid * PER + k;
      deg[k] = i < n ? deg[k] - rs[k] : 0;
      if (i < n) sm.start[i] = rs[k];
      local += deg[k];
    }
    int m;
    int ex = dev::block_exclusive_sum<TINY_THREADS>(local, sm.wave, &m);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid * PER + k;
      if (i < n) sm.seg[i] = ex;
      ex += deg[k];
    }
    if (tid == 0) { sm.seg[n] = m; sm.n = 0; }
    __syncthreads();
    const bool heavy = do_enabled && (long long)m > (n_edges_total - edges_before - edges_done) / 14 && n > 256;
    if (m > TINY_EDGES || heavy) {
      // ---- too big for one workgroup (or the direction might switch): hand the frontier back
      const int p = level & 1;
      if constexpr (policy_has_accept<Policy>::value)
        if (edges_done > 0 || vertices_done > 0)  // levels ran here: the frontier bitmaps are stale
          pol.tiny_hand_back(level, cur, n, CAP, a.frontier[p]);
      const int tiles = (n + TILE - 1) / TILE;
      for (int slot = tid; slot < tiles * TILE; slot += TINY_THREADS) a.frontier[p][slot] = slot < n ? cur[slot] : -1;
      if (tid < tiles) {
        const int lo = tid * TILE, hi = min(n, lo + TILE);
        const int sum = sm.seg[hi] - sm.seg[lo];
        a.tile_sums[tid] = sum;
        a.tile_chunks[tid] = (sum + CHUNK - 1) / CHUNK;
        a.tile_count[tid] = hi - lo;
      }
      if (tid == 0) {
        c->n_tiles[p] = tiles;
        c->level = level - 1;
        c->edges_visited = edges_before + edges_done;
        c->vertices_visited += vertices_done;
      }
      __syncthreads();  // a plan/decide step may follow in the same workgroup
      return 2;
    }
    // ---- the level itself -----------------------------------------------------------
    pol.set_level(level);

    int32_t* spill = a.frontier[(level + 1) & 1];  // next-frontier entries beyond CAP
    for (int base = 0; base < m; base += TINY_THREADS * TINY_ITEMS) {
      int e_k[TINY_ITEMS], slot_k[TINY_ITEMS], n_k[TINY_ITEMS], cand_k[TINY_ITEMS];
#pragma unroll
      for (int k = 0; k < TINY_ITEMS; ++k) {
        const int atom = base + k * TINY_THREADS + tid;
        int lo = 0;
        e_k[k] = -1;
        if (atom < m) {
#pragma unroll
          for (int step = SEARCH0; step >= 1; step >>= 1)
            if (lo + step < n && sm.seg[lo + step] <= atom) lo += step;
          e_k[k] = sm.start[lo] + (atom - sm.seg[lo]);
        }
        slot_k[k] = lo;
      }
#pragma unroll
      for (int k = 0; k < TINY_ITEMS; ++k) n_k[k] = a.ci[e_k[k] >= 0 ? e_k[k] : 0];  // m > 0: edge 0 exists
      bool pre_k[TINY_ITEMS];
#pragma unroll
      for (int k = 0; k < TINY_ITEMS; ++k) {
        const bool ok = e_k[k] >= 0;
        cand_k[k] = 0;
        typename Policy::src_state st{};
        if constexpr (!STATELESS) st = sm.state[slot_k[k]];
        const bool pass = pol.precheck(st, n_k[k], ok ? e_k[k] : 0, cand_k[k]);
        pre_k[k] = pass & ok;
      }
      int r1_k[TINY_ITEMS], r2_k[TINY_ITEMS];
#pragma unroll
      for (int k = 0; k < TINY_ITEMS; ++k) {
        r1_k[k] = 0;
        r2_k[k] = 0;
        if (pre_k[k]) r1_k[k] = pol.claim(n_k[k], cand_k[k]);
      }
      if constexpr (policy_two_claims<Policy>::value) {
#pragma unroll
        for (int k = 0; k < TINY_ITEMS; ++k) {
          const bool need = pre_k[k] & pol.need2(r1_k[k], cand_k[k]);
          if (need) r2_k[k] = pol.claim2(n_k[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < TINY_ITEMS; ++k) {
        int code = 0;
        if (pre_k[k]) code = pol.code(r1_k[k], r2_k[k], n_k[k], cand_k[k]);
        const bool keep = code == 1;
        const unsigned long long mk = dev::ballot(keep);
        if (mk) {
          int at = 0;
          if (lane == 0) at = atomicAdd(&sm.n, __popcll(mk));
          at = dev::wave_bcast0(at) + dev::mask_rank(mk);
          if (keep) {
            if (at < CAP) nxt[at] = n_k[k];
            else spill[at] = n_k[k];
            if constexpr (policy_has_accept<Policy>::value) pol.on_accept(n_k[k]);
          }
        }
      }
    }
    edges_done += m;
    vertices_done += n;
    __syncthreads();
    n = sm.n;
    sel ^= 1;
    ++level;
    __syncthreads();
  }
}

template <class Policy>
__global__ __launch_bounds__(TINY_THREADS) void tiny_levels_kernel(pipe_args a, Policy pol, int do_enabled,
                                                                   long long n_edges_total) {
  __shared__ tiny_smem<Policy> sm;
  (void)tiny_levels_body(a, pol, do_enabled, n_edges_total, sm, load_ctrl_head(a.ctrl));
}

}  // namespace grx
