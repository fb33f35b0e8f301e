// grx_bfs_kernels.hpp -- device pieces of BFS shared by the single-GPU enactor (grx_bfs.hip)
// and the partitioned multi-GPU enactor (grx_dist.hip): the claim policy, the Beamer
// switch constants and the bottom-up step.
#pragma once

#include "grx_frontier.hpp"

#include <climits>

namespace grx {

// element i (0..2) of a 3-array held in registers / kernel arguments WITHOUT dynamic indexing
// (which would move the whole enclosing struct to scratch memory)
template <class T>
__device__ __forceinline__ T* pick3(T* const (&arr)[3], int i) {
  return i == 0 ? arr[0] : (i == 1 ? arr[1] : arr[2]);
}

constexpr int BU_MORE = 1 << 30;  // dobfs_args::heads: flag in the second entry, "more than two in-edges"
// Beamer's alpha: a top-down level hands over to bottom-up when its frontier has more out-edges than 1 / alpha of the edges of
// the unexplored vertices.  14 is the paper's value for its kernels; here a bottom-up level costs 20-36 us on the LJ / deep
// stand-ins whatever the frontier, a claim-per-edge top-down level ~17 us per million edges: the break-even lies at a smaller
// frontier.  Measured, alpha = 14 / 28 / 56 / 112 (profiles/r5_c28_do_alpha.txt): 16 other sources of the LJ stand-in 253 / 270 /
// 266 / 243 GTEPS, deep stand-in 0.326 / 0.309 / 0.311 / 0.311 ms, hub source and twitter stand-in unchanged.  GRX_DO_ALPHA.
constexpr int DO_ALPHA = 28;
constexpr int DO_BETA = 24;

// VARIANT 0: the reference's claim, atomicMin on the label array with a stale
// pre-check (fastest measured).  Tuning variants for A/B runs (engine_flags bits 8-10):
// 1 = claim on a visited bitmap (atomicOr), 2 = same with an agent-scope pre-check,
// 3 = variant 1 counting attempted atomics in ctrl->spare[0], 7 = variant 0 on the
// stand-alone plan + advance kernels.
// Tried and rejected (round 1, LJ stand-in, top-down only): PER-XCD FILTER BITMAPS -- each
// XCD keeps a private V-bit "already tried" bitmap in its own L2 (workgroup-scope atomicOr,
// XCD id from HW_REG_XCC_ID) and only the first edge per (XCD, vertex) goes on to the label.
// Correct, but the fat levels got SLOWER (31 M-edge level: 496 -> 951 us; plain or sc1
// bitmap loads alike): one more dependent L2 round trip and one more atomic per surviving edge
// cost more than the label probes they save.
template <int VARIANT>
struct bfs_policy_t {
  using src_state = int;
  static constexpr bool stateless = true;  // load_source() carries nothing
  int32_t* dist;
  unsigned* visited;
  int next_depth;
  ctrl_t* ctrl;
  // direction-optimising runs (variant 0): the top-down levels keep the bitmap view of the
  // search up to date, so a switch to bottom-up needs no conversion pass -- every discovery sets
  // its bit in the NEXT frontier bitmap (one fire-and-forget atomic per discovered vertex; top-down
  // levels of such a run discover few), and the frontier being expanded is OR-ed into bm_visited.  bm_f[L % 3] is the frontier
  // of level L; level L writes bm_f[(L + 1) % 3] and clears bm_f[(L + 2) % 3] for level L + 1.
  unsigned* bm_visited;  // null: no bitmaps (forward-only runs)
  unsigned* bm_f[3];
  int bm_words;
  unsigned* bm_next;     // set by begin / set_level
  int in_tiny;           // set by tiny_levels_body: only bm_visited is maintained
  // forward-only runs with a visited bitmap (grx_bin.hpp): bm_f[0..2] all alias bm_visited, so
  // every discovery sets its bit there directly and nothing is ever cleared or folded
  int fwd_bitmap;
  // set by mid_levels_body (grx_mid.hpp): every workgroup of the launch sits on ONE XCD and nobody else touches
  // the labels, so the claim may execute in that XCD's L2 (workgroup scope) instead of at the memory side
  int l2_local;
  // read-only bitmap pre-filter of the label probe (null: off).  The bitmap is 32 x denser than
  // the labels (V / 8 bytes: L2-resident), and a neighbour it already shows as visited -- most
  // of them on the fat levels -- costs no label sector at all; it may lag behind (it is a hint,
  // the atomicMin on the label decides).
  const unsigned* pre_bm;

  __device__ __forceinline__ void set_level(int level) {
    next_depth = level + 1;
    if constexpr (VARIANT == 0)
      if (bm_visited) bm_next = pick3(bm_f, (level + 1) % 3);
  }
  __device__ __forceinline__ void begin(ctrl_t* c) { ctrl = c; set_level(c->level); }
  // a vertex joined the next frontier
  __device__ __forceinline__ void on_accept(int n) const {
    if constexpr (VARIANT == 0) {
      if (bm_visited) {
        const unsigned bit = 1u << (n & 31);
        // regular top-down level: ONE atomic (next frontier); the level kernel folds each frontier
        // bitmap into bm_visited with a streaming sweep when it is expanded.  Tiny levels: only
        // bm_visited (tiny_hand_back rebuilds the frontier bitmap).
        if (!in_tiny) (void)__hip_atomic_fetch_or(&bm_next[n >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else (void)__hip_atomic_fetch_or(&bm_visited[n >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // tiny_levels_body keeps only `bm_visited` current while it runs levels inside one workgroup
  // (nobody reads a frontier bitmap there); when it hands level `level` back to the regular
  // kernels it restores their invariant: bm_f[level % 3] = exactly the frontier (n vertices:
  // the first min(n, cap) in `lds`, the others at spill[cap ..], re-read past the L1),
  // bm_f[(level + 1) % 3] empty.  Two 0.6 MB clears by one workgroup: a few us, once per
  // hand-back, instead of bitmap upkeep in every tiny level.  Block-wide call.
  // entry: the frontier tiny levels start from (n vertices, in LDS) was produced by a regular level
  // and would have been folded into bm_visited by the regular level kernel expanding it
  __device__ __forceinline__ void tiny_enter(const int* lds, int n) {
    in_tiny = 1;
    if constexpr (VARIANT == 0) {
      if (!bm_visited) return;
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int v = lds[i];
        (void)__hip_atomic_fetch_or(&bm_visited[v >> 5], 1u << (v & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __device__ __forceinline__ void tiny_hand_back(int level, const int* lds, int n, int cap, const int* spill) {
    if constexpr (VARIANT == 0) {
      if (!bm_visited || fwd_bitmap) return;
      uint4* f0 = reinterpret_cast<uint4*>(pick3(bm_f, level % 3));
      uint4* f1 = reinterpret_cast<uint4*>(pick3(bm_f, (level + 1) % 3));
      for (int w = threadIdx.x; w < bm_words / 4; w += blockDim.x) {
        f0[w] = make_uint4(0u, 0u, 0u, 0u);
        f1[w] = make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
      unsigned* f = pick3(bm_f, level % 3);
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int v = i < cap ? lds[i] : __hip_atomic_load(&spill[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_or(&f[v >> 5], 1u << (v & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
  }
  __device__ __forceinline__ src_state load_source(int) const { return 0; }
  __device__ __forceinline__ bool precheck(src_state, int n, int, int&) const {
    if constexpr (VARIANT == 0) {
      if (pre_bm) {
        // the label load stays unconditional (a visited neighbour reads label 0: one broadcast
        // line), so the probes of a lane's edges still travel together -- see WHY PHASES
        const bool vis = (pre_bm[n >> 5] >> (n & 31)) & 1u;
        const int d = dist[vis ? 0 : n];
        return !vis && d > next_depth;
      }
    }
    if constexpr (VARIANT == 0 || VARIANT == 7) return dist[n] > next_depth;
    if constexpr (VARIANT == 2)
      return (__hip_atomic_load(&visited[n >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << (n & 31))) == 0u;
    return (visited[n >> 5] & (1u << (n & 31))) == 0u;
  }
  // what precheck computes, without the probe (mid_levels_body)
  __device__ __forceinline__ bool prepare(src_state, int, int, int&) const { return true; }
  __device__ __forceinline__ int claim(int n, int) const {
    if constexpr (VARIANT == 0) {
      if (l2_local) return __hip_atomic_fetch_min(&dist[n], next_depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if constexpr (VARIANT == 0 || VARIANT == 7) return atomicMin(&dist[n], next_depth);
    if constexpr (VARIANT == 3) atomicAdd(&ctrl->spare[0], 1);
    return (int)atomicOr(&visited[n >> 5], 1u << (n & 31));
  }
  __device__ __forceinline__ int code(int raw, int, int n, int) const {
    if constexpr (VARIANT == 0 || VARIANT == 7) return next_depth < raw ? 1 : 0;
    if ((unsigned)raw & (1u << (n & 31))) return 0;
    dist[n] = next_depth;
    return 1;
  }
  // the whole chain for one precheck survivor
  __device__ __forceinline__ int visit(src_state, int n, int) const { return code(claim(n, 0), 0, n, 0); }
};
using bfs_policy = bfs_policy_t<0>;

// ---------------------------------------------------------------------------------------------------------------
// PARTITIONED searches (round 6).  The reference has no multi-GPU path at all
// (operators/advance/advance.hxx:129-132 throws for context.size() != 1); here rank r of P owns the vertex slice
// [lo, hi) = [r * S, min((r + 1) * S, V)), S a multiple of 2048, with the OUT-rows of that slice as a graph of the whole
// vertex range (rows of other ranks' vertices are empty; column ids are global) -- and the single-GPU engine's own
// bodies run on that graph: the claim-per-edge advance, the binned scatter + sweep pair, the second bottom-up body.
// What a partition adds is what happens to a target the rank does not own: it is reported ONCE to its owner through
// the outgoing bitmap `send` (P slices of S bits = one V-bit bitmap in global vertex order), deduplicated by `sent`
// (forward runs: the visited bitmap itself -- a remote vertex's bit there means "reported").  After the exchange
// bfs_part_post_kernel claims what the peers reported in this rank's slice and emits it as tiles like any other
// producer of a frontier.
struct part_args {
  int32_t P, rank;
  int32_t lo, hi;                // owned vertices
  int32_t slice_words;           // S / 32
  int32_t do_run;                // this search may switch direction (three frontier bitmaps, `sent` of its own)
  unsigned* send;                // [P][slice_words]: bit v = "this rank discovered v at this level" (v not owned)
  const unsigned* recv;          // [P][slice_words]: slice j = what rank j reported about THIS rank's vertices (local word
                                 // index); on a bottom-up level: the whole-graph frontier bitmap, slice j from rank j
  unsigned* sent;                // V bits: remote vertices this rank has reported (never reported twice)
  long long* stats_local;        // {frontier vertices, frontier out-edges, 0, 0} of the NEXT level, this rank's share
  const long long* stats_global; // ... summed over the ranks by the all-reduce that follows every level group
  long long e_global;            // edges of the whole graph (Beamer's rule)
};

// The claim on a partitioned graph: owned targets exactly as bfs_policy (label probe, atomicMin, bitmap upkeep);
// targets of other ranks: `sent` probe -> atomicOr on `sent` -> the winner sets the bit in `send` (fire and forget).
// One load from a SELECTED address per phase, so the phases of advance_block keep their one round trip each.
struct bfs_policy_part : bfs_policy {
  int lo, hi;
  unsigned* sent;
  unsigned* send;
  __device__ __forceinline__ bool owned(int n) const { return n >= lo && n < hi; }
  __device__ __forceinline__ bool precheck(src_state, int n, int, int& cand) const {
    const bool own = owned(n);
    cand = own ? 0 : (int)(1u << (n & 31));
    const unsigned* p = own ? reinterpret_cast<const unsigned*>(dist + n) : sent + (n >> 5);
    const unsigned v = *p;
    return own ? (int)v > next_depth : (v & (unsigned)cand) == 0u;
  }
  __device__ __forceinline__ int claim(int n, int cand) const {
    if (cand == 0) return atomicMin(&dist[n], next_depth);
    return (int)atomicOr(&sent[n >> 5], (unsigned)cand);
  }
  __device__ __forceinline__ int code(int raw, int, int n, int cand) const {
    if (cand == 0) return next_depth < raw ? 1 : 0;
    if (((unsigned)raw & (unsigned)cand) == 0u)
      (void)__hip_atomic_fetch_or(&send[n >> 5], (unsigned)cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  __device__ __forceinline__ int visit(src_state, int n, int) const {
    const int cand = owned(n) ? 0 : (int)(1u << (n & 31));
    return code(claim(n, cand), 0, n, cand);
  }
};

struct dobfs_args {
  const int32_t* t_ro;   // in-edges (transpose; the CSR itself for symmetric graphs)
  const int32_t* t_ci;
  int32_t* dist;
  unsigned* visited;     // bitmap of closed vertices (discovered, or without in-edges)
  unsigned* fbits[3];    // frontier bitmaps: of level L at [L % 3] (rot3) or [L & 1] (partitioned runs)
  int32_t rot3;          // single GPU: three rotating bitmaps, all formats always maintained
  int32_t n_words;       // 32-bit words per bitmap (even)
  int32_t n_edges;
  int32_t enabled;       // direction optimisation on
  int32_t alpha, beta;   // Beamer switch: bottom-up when m_f > m_u / alpha, back when n_f < V / beta ...
  int32_t back_div;      // ... or when the frontier's out-edges m_f drop below E / back_div (0: rule off)
  long long* bu_part;    // per workgroup {found, out-degree sum, open, probes} of the last bottom-up launch
  int32_t bu_grid;       // workgroups of the bottom-up launch
  // partitioned runs (grx_dist.hip): this rank owns the 64-vertex chunks [ch_lo, ch_lo + n_words / 2);
  // visited / fbits are indexed by LOCAL chunk, labels / offsets / frontier probes by GLOBAL id
  int32_t ch_lo;               // 0 on a single GPU
  const unsigned* fin_global;  // partitioned runs: the whole-graph frontier bitmap a bottom-up level probes (assembled by the
                               // all-to-all in part_args::recv); null: the rank's own fbits[]
  uint32_t xcc_mask;           // hardware XCC ids of this device (grx_mid.hpp)
  // single GPU, per graph: {first, second in-neighbour} of every vertex (-1: none), so that the first probe group of a
  // bottom-up level reads ONE coalesced 8-byte stream instead of two column indices per lane from 64 different rows
  const int2* heads;           // null: probe through t_ci
  // tuning aid (GRX_BU_DEBUG=<level>): per-wave phase clocks of the second bottom-up body at that level, 8 words per wave
  long long* debug;
  int32_t debug_level;
  int32_t thin_div, thin_min;  // thin top-down levels: thin_workgroups() of the launch take chunks (0: all of them)
};

// WORKGROUPS OF A THIN CLAIM-PER-EDGE LEVEL (round 5).  The level kernels are launched with the resident grid (4 workgroups per
// CU) whatever the level holds; every workgroup that discovers anything ends with a tile reservation on one control word and
// leaves a short tile behind, which the next head has to walk and the next level stages as a chunk of its own.  Measured
// (profiles/r5_c32_thin_level_workgroups.txt, r5_c33_*), workgroups -> us: a level of ~1300 chunks 1024 -> 23, 512 -> 18;
// 1940 chunks 1024 -> 33, 768 -> 29, 256 -> 41; ~2700 chunks 1024 -> 43, 512 -> 50 -- about 2.5 chunks per workgroup -- but a level
// of ~650 chunks that discovers next to nothing 1024 -> 8, 256 -> 14: one chunk per workgroup.  So: one chunk per workgroup up
// to 768 chunks, 2.5 from 1280 on, linear in between; the other workgroups of the launch leave at once.  div == 0: off.
__device__ __forceinline__ int thin_workgroups(int total_chunks, int grid, int div, int min_wgs) {
  (void)min_wgs;
  if (div <= 0 || total_chunks < 0) return grid;
  int per10 = 10;  // chunks per workgroup, times ten
  if (total_chunks >= 1280) per10 = 25;
  else if (total_chunks > 768) per10 = 10 + 15 * (total_chunks - 768) / 512;
  const int want = (int)(((long long)total_chunks * 10 + per10 - 1) / per10);
  return min(grid, max(1, want));
}

// Bottom-up level.  A wave owns 64 consecutive vertices (one "chunk") and works on
// BATCH chunks at a time so that the dependent load chain (visited word -> in-offsets
// -> in-neighbour -> frontier word) of several chunks is in flight together.
// `visited` already counts vertices without in-edges as closed.
// QUEUE: the discoveries are ALSO emitted as frontier tiles (wave-private LDS staging, one
// tile per 256 discoveries of a wave, leftovers merged per workgroup), so a following top-down
// level needs no bitmap -> queue conversion pass.  Tile indices come from a STATIC range per
// workgroup -- workgroup b owns [b * T, (b + 1) * T), T = tiles its vertices could fill + 1 --
// handed out through an LDS counter; unused indices are written as empty tiles.  (Reserving
// them with atomics on the shared tile counter, as the top-down kernel does, made a fat bottom-up
// level 35 us slower: 6144 waves hit one word that sustains ~90 atomics/us.)
template <int BATCH = 4, bool QUEUE = false>
struct bottomup_smem {
  static constexpr int STAGE = QUEUE ? TILE + 64 * BATCH : 1;  // per wave
  int cnt[ADV_BLOCK / 64];
  long long deg[ADV_BLOCK / 64];
  int open[ADV_BLOCK / 64];
  long long probe[ADV_BLOCK / 64];
  int sv[ADV_BLOCK / 64][STAGE];   // staged discoveries (vertex ids) of each wave
  int sd[ADV_BLOCK / 64][STAGE];   // ... and their out-degrees
  int bv[QUEUE ? ADV_BLOCK / 64 * TILE : 1];  // leftovers of the four waves, merged at the end
  int bd[QUEUE ? ADV_BLOCK / 64 * TILE : 1];
  int bcnt;
  int tix_next;  // QUEUE: tiles of the workgroup's static range handed out so far
  int tiles_out; // QUEUE: non-empty tiles of this workgroup (for its bu_part record)
};

// One tile (index tix) of `n` (<= TILE) staged vertices sv[lo ..] with known degrees sd[lo ..],
// written by ONE wave.
__device__ __forceinline__ void wave_emit_tile(const pipe_args& a, int q, int tix, const int* sv, const int* sd,
                                               int lo, int n) {
  const int lane = dev::lane_id();
  int dsum = 0;
#pragma unroll
  for (int i = 0; i < TILE / 64; ++i) {
    const int k = i * 64 + lane;
    const bool ok = k < n;
    a.frontier[q][(size_t)tix * TILE + k] = ok ? sv[lo + k] : -1;
    dsum += ok ? sd[lo + k] : 0;
  }
  dsum = dev::wave_sum(dsum);
  if (lane == 0) {
    a.tile_sums[tix] = dsum;
    a.tile_chunks[tix] = (dsum + CHUNK - 1) / CHUNK;
    a.tile_count[tix] = n;
  }
}

template <int BATCH = 4, bool QUEUE = false>
__device__ __forceinline__ void bfs_bottomup_block(const pipe_args& a, const dobfs_args& d, ctrl_t* c,
                                                   bottomup_smem<BATCH, QUEUE>& sm) {
  int* s_cnt = sm.cnt;
  long long* s_deg = sm.deg;
  int* s_open = sm.open;
  long long* s_probe = sm.probe;
  const int level = c->level;
  const int p = level & 1;
  const unsigned* __restrict__ fin = d.fin_global ? d.fin_global : pick3(d.fbits, d.rot3 ? level % 3 : p);
  unsigned* fout = pick3(d.fbits, d.rot3 ? (level + 1) % 3 : (p ^ 1));
  int wcnt = 0;  // QUEUE: staged discoveries of this wave
  // QUEUE: static tile range of this workgroup (see above); every wave runs `iters` rounds of BATCH chunks
  const int n_chunks_all = d.n_words / 2;
  const int iters = (n_chunks_all + (int)gridDim.x * (ADV_BLOCK / 64) * BATCH - 1) / ((int)gridDim.x * (ADV_BLOCK / 64) * BATCH);
  const int tiles_per_wg = BATCH * iters + 1;
  const int tile_base = (int)blockIdx.x * tiles_per_wg;
  if constexpr (QUEUE) {
    if (threadIdx.x == 0) sm.tix_next = 0;
    __syncthreads();
  }
  const int vbase = d.ch_lo * 64;  // global id of local chunk 0
  const int lane = dev::lane_id();
  const int wid = threadIdx.x >> 6;
  const int wave = (blockIdx.x * ADV_BLOCK + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * ADV_BLOCK) >> 6;
  const int n_chunks = d.n_words / 2;
  const bool same_csr = d.t_ro == a.ro;  // symmetric graph: out-degree == in-degree
  int my_cnt = 0;
  long long my_deg = 0;
  long long my_probes = 0;  // in-edges actually read (roofline accounting)
  int my_open = 0;
  constexpr int SERIAL = 8;
  // PREFILTER (single GPU): late bottom-up levels find most 64-vertex chunks completely closed, and
  // walking them round by round costs one dependent round trip per round for nothing.  Lane l looks
  // at chunk slot l of this wave (round l / BATCH, chunk l % BATCH) -- ONE round trip for all rounds
  // -- and the ballot tells which rounds have an open vertex at all.  Closed chunks need no output:
  // the next-frontier bitmap was cleared by the previous level's sweep.
  unsigned long long live_slots = ~0ull;
  if (d.rot3 && iters * BATCH <= 64) {
    const int r = lane / BATCH, j = lane % BATCH;
    const int ch = wave * BATCH + r * n_waves * BATCH + j;
    bool live = false;
    if (r < iters && ch < n_chunks) {
      const unsigned long long v = *reinterpret_cast<const unsigned long long*>(d.visited + 2 * ch) |
                                   *reinterpret_cast<const unsigned long long*>(fin + 2 * ch);
      unsigned long long in_range = ~0ull;  // lanes of the last chunk beyond V are never open
      const long long first = (long long)vbase + (long long)ch * 64;
      if (first + 64 > (long long)a.V) in_range = first >= (long long)a.V ? 0ull : ((1ull << ((long long)a.V - first)) - 1ull);
      // live: an open vertex, OR frontier bits that `visited` does not hold yet (they are merged
      // and written back by the round below)
      const unsigned long long vm = *reinterpret_cast<const unsigned long long*>(d.visited + 2 * ch);
      live = (~v & in_range) != 0ull || (v != vm);
    }
    live_slots = dev::ballot(live);
  }
  int round = -1;
  for (int ch0 = wave * BATCH; ch0 < n_chunks; ch0 += n_waves * BATCH) {
    ++round;
    if (round * BATCH < 64 && ((live_slots >> (round * BATCH)) & ((1ull << BATCH) - 1ull)) == 0ull) continue;
    unsigned long long vis[BATCH], vis_mem[BATCH];
    int b[BATCH], e[BATCH], odeg[BATCH];
    bool open[BATCH], found[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int ch = ch0 + j;
      vis[j] = ~0ull;
      vis_mem[j] = ~0ull;
      if (ch < n_chunks) {
        vis[j] = (unsigned long long)d.visited[2 * ch] | ((unsigned long long)d.visited[2 * ch + 1] << 32);
        vis_mem[j] = vis[j];
        // single GPU: the frontier of THIS level may not be in `visited` yet (the top-down levels
        // fold a frontier in when they expand it): its own words are at hand -- and they are
        // written back below even if this chunk discovers nothing, or the next bottom-up level
        // would take these vertices for unvisited
        if (d.rot3) vis[j] |= (unsigned long long)fin[2 * ch] | ((unsigned long long)fin[2 * ch + 1] << 32);
      }
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int v = vbase + (ch0 + j) * 64 + lane;
      open[j] = v < a.V && !((vis[j] >> lane) & 1ull);
      found[j] = false;
      b[j] = e[j] = odeg[j] = 0;
      if (open[j]) {
        b[j] = d.t_ro[v];
        e[j] = d.t_ro[v + 1];
        ++my_open;
      }
    }
    // the first two in-neighbours of every vertex of the chunks, issued TOGETHER with the row offsets (both depend only
    // on the vertex id): the first probe group then waits for one round trip, not for offsets -> column indices
    int h0[BATCH], h1[BATCH];
    const bool use_heads = d.heads != nullptr;  // uniform
    if (use_heads) {
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int v = vbase + (ch0 + j) * 64 + lane;
        const int2 h = d.heads[v < a.V ? v : a.V - 1];  // lanes of a wave read 512 consecutive bytes
        h0[j] = h.x;
        h1[j] = h.y & ~BU_MORE;  // (-1 stays negative; it is never used as an index)
      }
    }
    // phase A: up to SERIAL probes per lane, the BATCH chunks advance in lock step.
    // Probes are issued SPECULATIVELY in groups (2, 2, 4): a group's column indices are
    // loaded together, then its frontier words together -- two dependent round trips per
    // group instead of two per probe (the wave waits for its slowest lane, which nearly
    // always runs all SERIAL probes).  A group's extra loads fall in the cache line its
    // first probe fetches anyway.
    {
      auto probe_group = [&](auto r0_c, auto n_c) -> bool {
        constexpr int R0 = decltype(r0_c)::value, N = decltype(n_c)::value;
        int u[BATCH][N];
        bool act[BATCH][N];
        bool any = false;
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const bool live = open[j] && !found[j];
#pragma unroll
          for (int q = 0; q < N; ++q) {
            act[j][q] = live && b[j] + R0 + q < e[j];
            any |= act[j][q];
          }
        }
        if (dev::ballot(any) == 0ull) return false;
        // UNCONDITIONAL loads from a clamped index (some lane has an in-edge, so entry 0 exists):
        // a predicated load sits in its own basic block and the compiler then waits for each
        // one before issuing the next -- 2 * BATCH * N serialized round trips instead of 2
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
#pragma unroll
          for (int q = 0; q < N; ++q) u[j][q] = d.t_ci[act[j][q] ? b[j] + R0 + q : 0];
        unsigned w[BATCH][N];
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
#pragma unroll
          for (int q = 0; q < N; ++q) w[j][q] = fin[(act[j][q] ? u[j][q] : 0) >> 5];
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
#pragma unroll
          for (int q = 0; q < N; ++q) {
            if (act[j][q]) {
              ++my_probes;
              if (w[j][q] & (1u << (u[j][q] & 31))) found[j] = true;
            }
          }
        return true;
      };
      using std::integral_constant;
      static_assert(SERIAL == 8, "probe groups cover 8 probes");
      bool more;
      if (use_heads) {
        // first group (2 probes) from the dense array: same rule as probe_group<0, 2>, written out so that h0 / h1 stay in
        // registers (captured by the generic lambda they ended up on the stack, with a wait right behind their load)
        bool any = false;
        unsigned w0[BATCH], w1[BATCH];
        bool a0[BATCH], a1[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          a0[j] = open[j] && b[j] < e[j];
          a1[j] = open[j] && b[j] + 1 < e[j];
          any |= a0[j];
        }
        more = dev::ballot(any) != 0ull;
        if (more) {
#pragma unroll
          for (int j = 0; j < BATCH; ++j) {
            w0[j] = fin[(a0[j] ? h0[j] : 0) >> 5];
            w1[j] = fin[(a1[j] ? h1[j] : 0) >> 5];
          }
#pragma unroll
          for (int j = 0; j < BATCH; ++j) {
            if (a0[j]) {
              ++my_probes;
              if (w0[j] & (1u << (h0[j] & 31))) found[j] = true;
            }
            if (a1[j]) {
              ++my_probes;
              if (w1[j] & (1u << (h1[j] & 31))) found[j] = true;
            }
          }
        }
      } else {
        more = probe_group(integral_constant<int, 0>{}, integral_constant<int, 2>{});
      }
      if (more)
        if (probe_group(integral_constant<int, 2>{}, integral_constant<int, 2>{}))
          (void)probe_group(integral_constant<int, 4>{}, integral_constant<int, 4>{});
    }
    // phase B: long in-lists are scanned by the whole wave, 64 edges per step
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      unsigned long long pend = dev::ballot(open[j] && !found[j] && e[j] > b[j] + SERIAL);
      while (pend) {
        const int src_lane = __builtin_ctzll(pend);
        pend &= pend - 1;
        const int bb = __shfl(b[j], src_lane, 64) + SERIAL, ee = __shfl(e[j], src_lane, 64);
        bool hit = false;
        for (int k = bb; k < ee; k += 64) {
          const int kk = k + lane;
          bool h = false;
          if (kk < ee) {
            const int u = d.t_ci[kk];
            ++my_probes;
            h = (fin[u >> 5] & (1u << (u & 31))) != 0u;
          }
          if (dev::ballot(h)) { hit = true; break; }
        }
        if (lane == src_lane) found[j] = hit;
      }
    }
    // out-degrees (next level's edge count) of the DISCOVERED vertices only, all chunks in one
    // round trip; the other lanes read row 0 (one broadcast line).  Loading them up front for
    // every open vertex cost a second 4 V-byte stream per bottom-up level on directed graphs.
    if (!same_csr) {
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int v = found[j] ? vbase + (ch0 + j) * 64 + lane : 0;
        odeg[j] = a.ro[v + 1] - a.ro[v];
      }
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int ch = ch0 + j;
      if (ch >= n_chunks) continue;
      const unsigned long long nw = dev::ballot(found[j]);
      if (lane == 0) {
        fout[2 * ch] = (unsigned)nw;
        fout[2 * ch + 1] = (unsigned)(nw >> 32);
        const unsigned long long nv = vis[j] | nw;
        if (nv != vis_mem[j]) {
          d.visited[2 * ch] = (unsigned)nv;
          d.visited[2 * ch + 1] = (unsigned)(nv >> 32);
        }
      }
      if (found[j]) {
        d.dist[vbase + ch * 64 + lane] = level + 1;
        my_cnt += 1;
        my_deg += same_csr ? (e[j] - b[j]) : odeg[j];
      }
      if constexpr (QUEUE) {
        if (nw) {
          if (found[j]) {
            const int at = wcnt + dev::mask_rank(nw);
            sm.sv[wid][at] = vbase + ch * 64 + lane;
            sm.sd[wid][at] = same_csr ? (e[j] - b[j]) : odeg[j];
          }
          wcnt += __popcll(nw);
        }
      }
    }
    if constexpr (QUEUE) {
      if (wcnt >= TILE) {  // at most TILE - 1 + 64 * BATCH staged: one tile leaves, from the end
        int k = 0;
        if (lane == 0) k = atomicAdd(&sm.tix_next, 1);
        k = dev::wave_bcast0(k);
        wave_emit_tile(a, p ^ 1, tile_base + k, sm.sv[wid], sm.sd[wid], wcnt - TILE, TILE);
        wcnt -= TILE;
      }
    }
  }
  if constexpr (QUEUE) {
    // leftovers (< TILE per wave) of the four waves -> as few tiles as possible
    if (threadIdx.x == 0) sm.bcnt = 0;
    __syncthreads();
    int at = 0;
    if (lane == 0 && wcnt) at = atomicAdd(&sm.bcnt, wcnt);
    at = dev::wave_bcast0(at);
    for (int i = lane; i < wcnt; i += 64) {
      sm.bv[at + i] = sm.sv[wid][i];
      sm.bd[at + i] = sm.sd[wid][i];
    }
    __syncthreads();
    const int total = sm.bcnt;
    const int used = sm.tix_next;  // full tiles emitted during the sweep
    const int extra = (total + TILE - 1) / TILE;
    if (wid < extra) wave_emit_tile(a, p ^ 1, tile_base + used + wid, sm.bv, sm.bd, wid * TILE, min(TILE, total - wid * TILE));
    // the rest of the static range: empty tiles
    for (int t = used + extra + (int)threadIdx.x; t < tiles_per_wg; t += ADV_BLOCK) {
      a.tile_sums[tile_base + t] = 0;
      a.tile_chunks[tile_base + t] = 0;
      a.tile_count[tile_base + t] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      c->n_tiles[p ^ 1] = (int)gridDim.x * tiles_per_wg;
      c->bu_R = (int)gridDim.x;
      c->bu_T = tiles_per_wg;
    }
    if (threadIdx.x == 0) sm.tiles_out = used + extra;  // read after the barrier of the totals below
  }
  // per-workgroup totals -> a handful of atomics per workgroup
  my_cnt = dev::wave_sum(my_cnt);
  my_open = dev::wave_sum(my_open);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    my_deg += __shfl_xor(my_deg, o, 64);
    my_probes += __shfl_xor(my_probes, o, 64);
  }
  if (lane == 0) { s_cnt[wid] = my_cnt; s_deg[wid] = my_deg; s_open[wid] = my_open; s_probe[wid] = my_probes; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tc = 0, to = 0;
    long long td = 0, tp = 0;
#pragma unroll
    for (int i = 0; i < ADV_BLOCK / 64; ++i) { tc += s_cnt[i]; td += s_deg[i]; to += s_open[i]; tp += s_probe[i]; }
    // plain stores; the head kernel of the next level reduces them
    // word 0: discoveries (low 40 bits) | non-empty tiles of this workgroup's static range (QUEUE)
    d.bu_part[4 * blockIdx.x] = (long long)tc | (QUEUE ? ((long long)sm.tiles_out << 40) : 0ll);
    d.bu_part[4 * blockIdx.x + 1] = td;
    d.bu_part[4 * blockIdx.x + 2] = to;
    d.bu_part[4 * blockIdx.x + 3] = tp;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Bottom-up level, second version (round 3, single GPU): dense groups of OPEN vertices, one round trip per group.
//
// What the counters said about the first version (profiles/history/r3_bench_pmc.json, class bottom_up): 72 % of the wave cycles
// waiting, 80 G L2 requests/s, 1.5 TB/s -- neither bandwidth nor request rate.  A wave there walks, per round of BATCH
// 64-vertex chunks,  visited word -> row offsets (+ two first in-neighbours) -> frontier words  and then, for the FEW lanes
// that the first two probes did not settle (5-10 % of the open vertices), two more probe groups (column indices -> frontier
// words, twice), the whole-wave scan of long lists and the out-degrees of the discoveries: about nine dependent round trips
// and ~300 instructions per round, whatever the share of open vertices in the chunks -- 52 % on the first bottom-up level of
// the LJ stand-in, 10 % on the second, 0.2 % on the third.  Here:
//  * the visited|frontier words of ALL chunks ("slots") of a wave are read once (by the workgroup: 8 lanes share a line),
//    and kept one slot per lane (two per lane: 128 slots);
//  * the open vertices of the slots are COMPACTED into a wave-private LDS list ((slot, lane): 16 bits each) -- ballots and
//    LDS stores only -- and worked off in dense groups of 2 x 64: the third level of the LJ stand-in is one group per wave
//    instead of two rounds of mostly closed lanes, the second one or two instead of eight;
//  * a group needs no row offsets: the dense two-neighbour array says whether a vertex has a first / second / further
//    in-neighbour (BU_MORE); the entries of group g + 1 are requested before group g is worked on;
//  * a group probes only the two first in-neighbours.  Lanes that are still open and have more in-edges are DEFERRED to a
//    second wave-private list which is worked off 64 entries at a time with every lane busy (four column indices per lane as
//    one 16-byte load while many lanes are alive, then the whole wave per long list) -- at the end, or when the list is full;
//  * out-degrees of discoveries are read when a tile of 256 is emitted (four per lane, one round trip per tile);
//  * discovered bits are collected per slot in LDS and every changed frontier / visited word is written once, at the end,
//    one slot per lane (the words belong to this wave alone).
// Needs dobfs_args::heads and iters * BATCH <= 128 (else the host launches the first version).
template <int BATCH>
struct bottomup2_smem {
  static constexpr int NW = ADV_BLOCK / 64;
  static constexpr int GROUP = 64 * BATCH;         // open vertices worked on together
  static constexpr int STAGE = TILE + GROUP;       // discoveries staged per wave
  static constexpr int DEFER = 2 * GROUP;          // deferred entries per wave
  static constexpr int OPEN = 2 * GROUP + 64;      // compacted open vertices per wave
  static constexpr int SLOTS = 128;                // chunks of a wave
  int cnt[NW];
  long long deg[NW];
  int open[NW];
  long long probe[NW];
  int sv[NW][STAGE];
  union {
    struct {
      unsigned short dv[NW][DEFER];  // (slot << 6) | lane
      unsigned short ov[NW][OPEN];   // (slot << 6) | lane
    } q;
    int bv[NW * TILE];         // leftovers of the four waves, merged at the end (the lists are dead by then)
    unsigned long long pw[NW * SLOTS][2];  // prologue: {visited | frontier, visited} word of every slot of the workgroup
  } u;
  unsigned late[NW][2 * SLOTS];  // bits discovered in this launch, per slot
  int bcnt;
  int tix_next;
  int tiles_out;
};

// one tile (index tix) of n <= TILE staged vertices src[lo ..], written by ONE wave; out-degrees read here.  Returns their sum.
__device__ __forceinline__ int wave_emit_tile_deg(const pipe_args& a, int q, int tix, const int* src, int lo, int n) {
  const int lane = dev::lane_id();
  int v[TILE / 64], r0[TILE / 64], r1[TILE / 64];
#pragma unroll
  for (int i = 0; i < TILE / 64; ++i) {
    const int k = i * 64 + lane;
    v[i] = k < n ? src[lo + k] : -1;
  }
#pragma unroll
  for (int i = 0; i < TILE / 64; ++i) {  // unconditional loads from a clamped index: one round trip for all
    const unsigned vv = v[i] >= 0 ? (unsigned)v[i] : 0u;
    r0[i] = a.ro[vv];
    r1[i] = a.ro[vv + 1u];
  }
  int dsum = 0;
#pragma unroll
  for (int i = 0; i < TILE / 64; ++i) {
    a.frontier[q][(size_t)tix * TILE + i * 64 + lane] = v[i];
    dsum += v[i] >= 0 ? r1[i] - r0[i] : 0;
  }
  dsum = dev::wave_sum(dsum);
  if (lane == 0) {
    a.tile_sums[tix] = dsum;
    a.tile_chunks[tix] = (dsum + CHUNK - 1) / CHUNK;
    a.tile_count[tix] = n;
  }
  return dsum;
}

template <int BATCH, bool DBG = false>
__device__ __forceinline__ void bfs_bottomup2_block(const pipe_args& a, const dobfs_args& d, ctrl_t* c,
                                                    bottomup2_smem<BATCH>& sm) {
  using S = bottomup2_smem<BATCH>;
  static_assert(BATCH == 2, "groups of 2 x 64 open vertices");
  // tuning clocks (DBG builds only): 100 MHz wall clock; `settle` waits for every outstanding load first so that the
  // round trip is charged to the phase that ends there
  auto clk = [&](bool settle) -> long long {
    if constexpr (DBG) {
      if (settle) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("" ::: "memory");
      const long long t = (long long)wall_clock64();
      asm volatile("" ::: "memory");
      return t;
    }
    return 0ll;
  };
  long long t_pro = 0, t_probe = 0, t_out = 0, t_drain = 0, t_emit = 0, t_tail = 0, t_compact = 0;
  int n_rounds = 0, n_deferred = 0, n_drains = 0;
  const long long t_start = clk(false);
  const int level = c->level;
  const int p = level & 1;
  const unsigned* __restrict__ fin = d.fin_global ? d.fin_global : pick3(d.fbits, level % 3);
  unsigned* fout = pick3(d.fbits, (level + 1) % 3);
  const int n_chunks = d.n_words / 2;
  const int iters = (n_chunks + (int)gridDim.x * S::NW * BATCH - 1) / ((int)gridDim.x * S::NW * BATCH);
  const int tiles_per_wg = BATCH * iters + 1;
  const int tile_base = (int)blockIdx.x * tiles_per_wg;
  const int lane = dev::lane_id();
  const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wave = (int)blockIdx.x * S::NW + wid;
  const int n_waves = (int)gridDim.x * S::NW;
  int* sv = sm.sv[wid];
  unsigned short* dv = sm.u.q.dv[wid];
  unsigned short* ov = sm.u.q.ov[wid];
  unsigned* late = sm.late[wid];
  if (threadIdx.x == 0) sm.tix_next = 0;
#pragma unroll
  for (int i = 0; i < 2 * S::SLOTS / 64; ++i) late[i * 64 + lane] = 0u;

  // every slot of this wave: slot s = chunk s % BATCH of round s / BATCH; lane l looks at slots l and l + 64.
  // The words are fetched by the WORKGROUP: the four waves own 4 * BATCH consecutive chunks per round, so thread t takes
  // chunk t % (4 * BATCH) of round t / (4 * BATCH) and 8 lanes share a 64-byte line; exchanged through LDS.
  {
    constexpr int PER_ROUND = S::NW * BATCH;
#pragma unroll
    for (int k = 0; k < S::NW * S::SLOTS / ADV_BLOCK; ++k) {
      const int ws = (int)threadIdx.x + ADV_BLOCK * k;
      const int r = ws / PER_ROUND, cc = ws % PER_ROUND;
      const int ch = ((int)blockIdx.x * S::NW + r * n_waves) * BATCH + cc;
      unsigned long long v = ~0ull, vm = ~0ull;
      if (r < iters && ch < n_chunks) {
        vm = *reinterpret_cast<const unsigned long long*>(d.visited + 2 * ch);
        v = vm | *reinterpret_cast<const unsigned long long*>(fin + 2 * ch);
      }
      sm.u.pw[ws][0] = v;
      sm.u.pw[ws][1] = vm;
    }
  }
  __syncthreads();
  auto chunk_of = [&](int s) -> int { return (wave + (s / BATCH) * n_waves) * BATCH + s % BATCH; };
  // lane l keeps the words of slots l (pv0) and l + 64 (pv1), with the vertices beyond V counted as closed; a slot is
  // LIVE when it has an open vertex, DIRTY when its frontier bits are not in `visited` yet (top-down levels fold a frontier
  // in when they expand it: such words are written back even if the chunk discovers nothing)
  unsigned long long pv0 = ~0ull, pv1 = ~0ull, live0 = 0ull, live1 = 0ull;
  bool dirty0 = false, dirty1 = false;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int s = lane + 64 * k;
    const int ch = chunk_of(s);
    const int ws = (s / BATCH) * (S::NW * BATCH) + wid * BATCH + s % BATCH;
    unsigned long long v = sm.u.pw[ws][0];
    const unsigned long long vm = sm.u.pw[ws][1];
    const bool valid = s / BATCH < iters && ch < n_chunks;
    const bool dirty = valid && v != vm;
    const long long first = (long long)ch * 64;
    if (valid && first + 64 > (long long)a.V) v |= first >= (long long)a.V ? ~0ull : (~0ull << ((long long)a.V - first));
    if (!valid) v = ~0ull;
    if (k == 0) { pv0 = v; dirty0 = dirty; live0 = dev::ballot(~v != 0ull); }
    else { pv1 = v; dirty1 = dirty; live1 = dev::ballot(~v != 0ull); }
  }
  __syncthreads();  // (the lists overlay the exchange area)
  if constexpr (DBG) t_pro = clk(true) - t_start;

  int my_cnt = 0, my_open = 0, my_probes = 0;
  long long my_deg = 0;
  int wcnt = 0;  // staged discoveries of this wave
  int dcnt = 0;  // deferred entries of this wave
  int ocnt = 0;  // compacted open vertices of this wave
  const unsigned last_edge = d.n_edges > 0 ? (unsigned)d.n_edges - 1u : 0u;
  auto vertex_of = [&](unsigned ent) -> int { return chunk_of((int)(ent >> 6)) * 64 + (int)(ent & 63u); };

  auto emit_from_end = [&]() {  // one full tile leaves the staging area, from the end
    const long long te0 = clk(false);
    int k = 0;
    if (lane == 0) k = atomicAdd(&sm.tix_next, 1);
    k = dev::wave_bcast0(k);
    const int dsum = wave_emit_tile_deg(a, p ^ 1, tile_base + k, sv, wcnt - TILE, TILE);
    if (lane == 0) my_deg += dsum;
    wcnt -= TILE;
    if constexpr (DBG) t_emit += clk(false) - te0;
  };
  // a discovery (call with the ballot of `fnd` over the wave): label, bit, staging
  auto discovered = [&](bool fnd, unsigned ent, int v) {
    const unsigned long long nw = dev::ballot(fnd);
    if (nw) {
      if (fnd) {
        d.dist[(unsigned)v] = level + 1;
        my_cnt += 1;
        atomicOr(&late[2u * (ent >> 6) + ((ent >> 5) & 1u)], 1u << (ent & 31u));
        sv[wcnt + dev::mask_rank(nw)] = v;
      }
      wcnt += __popcll(nw);
    }
  };

  // the deferred pass: every lane takes one entry
  auto drain = [&]() {
    const long long td0 = clk(false);
    if constexpr (DBG) { n_deferred += dcnt; ++n_drains; }
    for (int base = 0; base < dcnt; base += 64) {
      const int i = base + lane;
      const bool act = i < dcnt;
      const unsigned ent = act ? (unsigned)dv[i] : 0u;
      const int v = vertex_of(ent);
      const unsigned rv = act ? (unsigned)v : 0u;
      const int rb = d.t_ro[rv], e = act ? d.t_ro[rv + 1u] : 0;  // (one 8-byte load)
      int pos = rb + 2;  // the two first in-neighbours were probed in the group
      bool fnd = false;
      int it = 0;
      for (;;) {
        // four column indices as ONE 16-byte load (one line request per lane instead of four); a list that ends within
        // three entries of the END OF THE ARRAY is left to the whole-wave scan below, which tests every index
        const bool go = act && !fnd && pos < e && (unsigned)pos + 3u <= last_edge;
        const unsigned long long mm = dev::ballot(go);
        if (mm == 0ull) break;
        if (it >= 2 && __popcll(mm) <= 8) break;  // a few long lists: the whole wave per list, below
        struct __attribute__((packed, aligned(4))) quad { int x[4]; };
        const quad u = *reinterpret_cast<const quad*>(d.t_ci + (go ? (unsigned)pos : 0u));
        unsigned w[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ok[q] = go && pos + q < e;
          w[q] = fin[(unsigned)(ok[q] ? u.x[q] : 0) >> 5];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          my_probes += ok[q] ? 1 : 0;
          fnd |= ok[q] && ((w[q] >> (u.x[q] & 31)) & 1u) != 0u;
        }
        pos += go ? 4 : 0;
        ++it;
      }
      unsigned long long pend = dev::ballot(act && !fnd && pos < e);
      while (pend) {
        const int src_lane = __builtin_ctzll(pend);
        pend &= pend - 1;
        const int bb = __shfl(pos, src_lane, 64), ee = __shfl(e, src_lane, 64);
        bool hit = false;
        for (int k = bb; k < ee; k += 64) {
          const int kk = k + lane;
          bool h = false;
          if (kk < ee) {
            const int u = d.t_ci[kk];
            ++my_probes;
            h = (fin[(unsigned)u >> 5] & (1u << (u & 31))) != 0u;
          }
          if (dev::ballot(h)) { hit = true; break; }
        }
        if (lane == src_lane) fnd = hit;
      }
      discovered(fnd, ent, v);
      if (wcnt >= TILE) emit_from_end();
    }
    dcnt = 0;
    if constexpr (DBG) t_drain += clk(true) - td0;
  };

  // live slots, in order: one 64-bit mask after the other
  unsigned long long todo0 = live0, todo1 = live1;
  constexpr unsigned NONE = 0xffffu;
  unsigned ne0 = NONE, ne1 = NONE, ce0 = NONE, ce1 = NONE;  // entries of the group in flight / of the current group
  int nh00 = -1, nh01 = -1, nh10 = -1, nh11 = -1;          // their two first in-neighbours
  int ch00 = -1, ch01 = -1, ch10 = -1, ch11 = -1;
  bool have_cur = false;
  for (;;) {
    // 1. compact open vertices until a group is full or the slots are used up
    const long long tc0 = clk(false);
    while (ocnt < S::GROUP && (todo0 | todo1) != 0ull) {
      int s;
      if (todo0) { s = (int)__builtin_ctzll(todo0); todo0 &= todo0 - 1ull; }
      else { s = 64 + (int)__builtin_ctzll(todo1); todo1 &= todo1 - 1ull; }
      const unsigned long long w = (s >> 6) ? pv1 : pv0;
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)w, s & 63);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(w >> 32), s & 63);
      const unsigned long long m = ~((unsigned long long)lo | ((unsigned long long)hi << 32));  // open lanes of slot s
      if ((m >> lane) & 1ull) ov[ocnt + dev::mask_rank(m)] = (unsigned short)((s << 6) | lane);
      ocnt += __popcll(m);
    }
    if constexpr (DBG) t_compact += clk(false) - tc0;
    // 2. the next group: its entries leave the list (from the end), its two-neighbour records are requested
    const int n_new = ocnt < S::GROUP ? ocnt : S::GROUP;
    const bool have_new = n_new > 0;
    if (have_new) {
      const int base = ocnt - n_new;
      ne0 = lane < n_new ? (unsigned)ov[base + lane] : NONE;
      ne1 = 64 + lane < n_new ? (unsigned)ov[base + 64 + lane] : NONE;
      ocnt = base;
      my_open += (ne0 != NONE ? 1 : 0) + (ne1 != NONE ? 1 : 0);
      const int2 h0 = d.heads[ne0 != NONE ? (unsigned)vertex_of(ne0) : 0u];  // (idle lanes read entry 0: one broadcast line)
      const int2 h1 = d.heads[ne1 != NONE ? (unsigned)vertex_of(ne1) : 0u];
      nh00 = h0.x; nh01 = h0.y; nh10 = h1.x; nh11 = h1.y;
    }
    // 3. the current group: frontier words of its first two in-neighbours
    if (have_cur) {
      if (dcnt + S::GROUP > S::DEFER) drain();
      const long long tp0 = clk(false);
      if constexpr (DBG) ++n_rounds;
      const bool o0 = ce0 != NONE, o1 = ce1 != NONE;
      const bool a00 = o0 && ch00 >= 0, a01 = o0 && ch01 >= 0, a10 = o1 && ch10 >= 0, a11 = o1 && ch11 >= 0;
      const unsigned u00 = a00 ? (unsigned)ch00 : 0u, u01 = a01 ? (unsigned)(ch01 & ~BU_MORE) : 0u;
      const unsigned u10 = a10 ? (unsigned)ch10 : 0u, u11 = a11 ? (unsigned)(ch11 & ~BU_MORE) : 0u;
      const unsigned w00 = fin[u00 >> 5], w01 = fin[u01 >> 5], w10 = fin[u10 >> 5], w11 = fin[u11 >> 5];
      // (bitwise: a short-circuit here put each frontier word's wait into a branch of its own)
      const bool f0 = (((w00 >> (u00 & 31u)) & (a00 ? 1u : 0u)) | ((w01 >> (u01 & 31u)) & (a01 ? 1u : 0u))) != 0u;
      const bool f1 = (((w10 >> (u10 & 31u)) & (a10 ? 1u : 0u)) | ((w11 >> (u11 & 31u)) & (a11 ? 1u : 0u))) != 0u;
      my_probes += (a00 ? 1 : 0) + (a01 ? 1 : 0) + (a10 ? 1 : 0) + (a11 ? 1 : 0);
      // still open, more in-edges: deferred.  Every ballot of the group is taken BEFORE its first store: the compiler
      // otherwise sinks the test of the second half behind the stores of the first, and its wait for the frontier words
      // then also waits for those stores (vmcnt retires in order)
      const bool df0 = a01 && (ch01 & BU_MORE) != 0 && !f0, df1 = a11 && (ch11 & BU_MORE) != 0 && !f1;
      const unsigned long long dm0 = dev::ballot(df0), dm1 = dev::ballot(df1);
      const long long tp1 = clk(true);
      if constexpr (DBG) t_probe += tp1 - tp0;
      if (dm0 | dm1) {
        if (df0) dv[dcnt + dev::mask_rank(dm0)] = (unsigned short)ce0;
        dcnt += __popcll(dm0);
        if (df1) dv[dcnt + dev::mask_rank(dm1)] = (unsigned short)ce1;
        dcnt += __popcll(dm1);
      }
      discovered(f0, ce0, f0 ? vertex_of(ce0) : 0);
      discovered(f1, ce1, f1 ? vertex_of(ce1) : 0);
      if constexpr (DBG) t_out += clk(false) - tp1;
      if (wcnt >= TILE) emit_from_end();
    }
    if (!have_new) break;
    ce0 = ne0; ce1 = ne1;
    ch00 = nh00; ch01 = nh01; ch10 = nh10; ch11 = nh11;
    have_cur = true;
  }
  if (dcnt > 0) drain();
  const long long t_loop_end = clk(true);
  // every changed word of this wave, one slot per lane: the next frontier (cleared by the previous level's sweep, so only
  // non-zero words are written) and `visited`
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int s = lane + 64 * k;
    const unsigned long long nw = (unsigned long long)late[2 * s] | ((unsigned long long)late[2 * s + 1] << 32);
    const bool dirty = k == 0 ? dirty0 : dirty1;
    if (nw != 0ull || dirty) {  // (an invalid slot is neither)
      const int ch = chunk_of(s);
      const long long first = (long long)ch * 64;
      unsigned long long keep = ~0ull;  // the bits beyond V were set above to close them: not part of `visited`
      if (first + 64 > (long long)a.V) keep = first >= (long long)a.V ? 0ull : (1ull << ((long long)a.V - first)) - 1ull;
      if (nw != 0ull) *reinterpret_cast<unsigned long long*>(fout + 2 * ch) = nw;
      *reinterpret_cast<unsigned long long*>(d.visited + 2 * ch) = ((k == 0 ? pv0 : pv1) & keep) | nw;
    }
  }
  // leftovers (< TILE per wave) of the four waves -> as few tiles as possible
  if (threadIdx.x == 0) sm.bcnt = 0;
  __syncthreads();  // (every wave is done with its lists: u.bv overlays them)
  int at = 0;
  if (lane == 0 && wcnt) at = atomicAdd(&sm.bcnt, wcnt);
  at = dev::wave_bcast0(at);
  for (int i = lane; i < wcnt; i += 64) sm.u.bv[at + i] = sv[i];
  __syncthreads();
  const int total = sm.bcnt;
  const int used = sm.tix_next;  // full tiles emitted during the sweep
  const int extra = (total + TILE - 1) / TILE;
  if (wid < extra) {
    const int dsum = wave_emit_tile_deg(a, p ^ 1, tile_base + used + wid, sm.u.bv, wid * TILE, min(TILE, total - wid * TILE));
    if (lane == 0) my_deg += dsum;
  }
  // the rest of the static range: empty tiles
  for (int t = used + extra + (int)threadIdx.x; t < tiles_per_wg; t += ADV_BLOCK) {
    a.tile_sums[tile_base + t] = 0;
    a.tile_chunks[tile_base + t] = 0;
    a.tile_count[tile_base + t] = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    c->n_tiles[p ^ 1] = (int)gridDim.x * tiles_per_wg;
    c->bu_R = (int)gridDim.x;
    c->bu_T = tiles_per_wg;
  }
  if (threadIdx.x == 0) sm.tiles_out = used + extra;  // read after the barrier of the totals below
  // per-workgroup totals
  my_cnt = dev::wave_sum(my_cnt);
  my_open = dev::wave_sum(my_open);
  my_probes = dev::wave_sum(my_probes);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) my_deg += __shfl_xor(my_deg, o, 64);
  if (lane == 0) { sm.cnt[wid] = my_cnt; sm.deg[wid] = my_deg; sm.open[wid] = my_open; sm.probe[wid] = my_probes; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tc = 0, to = 0;
    long long td = 0, tp = 0;
#pragma unroll
    for (int i = 0; i < S::NW; ++i) { tc += sm.cnt[i]; td += sm.deg[i]; to += sm.open[i]; tp += sm.probe[i]; }
    d.bu_part[4 * blockIdx.x] = (long long)tc | ((long long)sm.tiles_out << 40);
    d.bu_part[4 * blockIdx.x + 1] = td;
    d.bu_part[4 * blockIdx.x + 2] = to;
    d.bu_part[4 * blockIdx.x + 3] = tp;
  }
  if constexpr (DBG) {
    const long long t_end = clk(true);
    t_tail = t_end - t_loop_end;
    if (lane == 0 && d.debug && level == d.debug_level && wave < 16384) {
      long long* o = d.debug + 8 * (size_t)wave;
      o[0] = (long long)blockIdx.x | (t_compact << 32);
      o[1] = t_start;
      o[2] = t_end;
      o[3] = t_pro | (t_tail << 32);
      o[4] = t_probe | (t_out << 32);
      o[5] = t_drain | (t_emit << 32);
      o[6] = (long long)n_rounds | ((long long)n_deferred << 16) | ((long long)n_drains << 40);
      o[7] = -3;
    }
  }
}

// Expand bitmap words into frontier tiles of parity q: thread t of the workgroup takes word
// base + t per round (word_of(w) -> 32 membership bits of vertices vbase + 32 w ..), the
// set bits are compacted through LDS (popcount + block scan) and flushed as tiles.
// Workgroup-wide call; the kernel's grid strides over [0, n_words).
struct words_smem {
  int out[TILE + ADV_BLOCK * 32];
  int wave[ADV_BLOCK / 64 + 1];
  int res[3];
  int cnt;
};

template <class WordFn>
__device__ __forceinline__ void words_to_tiles(const pipe_args& a, ctrl_t* c, int q, int n_words, int vbase,
                                               WordFn word_of, words_smem& sm) {
  const int tid = threadIdx.x;
  if (tid == 0) { sm.cnt = 0; sm.res[0] = 0; sm.res[1] = 0; }
  __syncthreads();
  for (int base = blockIdx.x * ADV_BLOCK; base < n_words; base += gridDim.x * ADV_BLOCK) {
    const int w = base + tid;
    unsigned bits = w < n_words ? word_of(w) : 0u;
    int cnt = __popc(bits);
    int tot;
    int at = sm.cnt + dev::block_exclusive_sum<ADV_BLOCK>(cnt, sm.wave, &tot);
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      sm.out[at++] = vbase + w * 32 + b;
    }
    __syncthreads();
    int have = sm.cnt + tot;
    __syncthreads();
    while (have >= TILE) {
      emit_tile(a, c, q, sm.out, have - TILE, TILE, sm.wave, sm.res);
      have -= TILE;
      __syncthreads();
    }
    if (tid == 0) sm.cnt = have;
    __syncthreads();
  }
  const int rem = sm.cnt;
  if (rem > 0) emit_tile(a, c, q, sm.out, 0, rem, sm.wave, sm.res);
  __syncthreads();
  release_tiles(a, sm.res);
}

}  // namespace grx
