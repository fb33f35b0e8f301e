// grx_bfs.hip -- breadth-first search on the device-driven frontier pipeline.
//
// Reference behaviour reproduced (include/gunrock/algorithms/bfs.hxx):
//   reset   :59-69   distances = INT_MAX, distances[source] = 0
//   advance :105-128 old = atomicMin(&dist[nbr], iteration + 1); keep if improved
//   filter  :130-146 drop the invalid (-1) slots
// Result parity: depths are bit-identical (the first-discovery level is unique).
//
// MI355X implementation
//  * top-down levels: advance_kernel<bfs_policy> -- stale pre-check on the label,
//    then the reference's atomicMin claim; exactly one thread wins a vertex and
//    emits it, so advance and filter are one kernel and no -1 reaches HBM.
//  * bottom-up levels (advance_direction = optimized; the reference declares the
//    enum, operators/configs.hxx:78-82, and ignores it): every unvisited vertex
//    scans its IN-edges until it finds a parent in the frontier bitmap.  A wave
//    owns 64 consecutive vertices, so bitmap words are assembled with a ballot and
//    written with plain stores -- no atomics at all; the frontier bitmap is
//    read-only during the level and stays L2 resident.  The switch follows Beamer's
//    heuristic (frontier edges vs unexplored edges / alpha; frontier size vs V / beta)
//    and is taken ON THE DEVICE by the head kernel, so the host still enqueues levels blindly.
//  * BOTH frontier formats are kept up to date all the time (queue of tiles + three rotating
//    bitmaps + the visited bitmap): top-down discoveries set their bits (two fire-and-forget
//    atomics per discovered vertex), bottom-up discoveries are also emitted as tiles.  A direction
//    switch therefore needs no conversion pass, and a level is exactly two launches (head + level).
//    Measured before this change (rocprofv3 kernel trace, LJ stand-in): the separate conversion
//    kernel cost 52 of the 250 us of a search -- 19 us labels -> bitmaps, 10 us bitmap -> queue,
//    and 4.5 us per level as a no-op launch the host had to enqueue blindly.
#include "grx_engine.hpp"
#include "grx_bfs_kernels.hpp"
#include "grx_bin.hpp"
#include "grx_mid.hpp"

#include <climits>
#include <cstddef>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace grx {

// problem.reset() of a direction-optimising run, ONE launch: labels = INT_MAX; visited = the
// graph's static "no in-edges" bitmap; frontier bitmaps 0 and 1 empty (2 is cleared by level 0).
// src >= 0: the source's label and bits are part of the fill (bfs_reset_seed_kernel below).
__device__ __forceinline__ void bfs_reset_body(int32_t* dist, int64_t V, const dobfs_args& d, const unsigned* closed0, int src) {
  const int64_t gsz = (int64_t)gridDim.x * blockDim.x, gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t sw = src >= 0 ? (src >> 5) : -1;
  const unsigned sbit = src >= 0 ? (1u << (src & 31)) : 0u;
  for (int64_t i = gid; i < V; i += gsz) dist[i] = i == (int64_t)src ? 0 : INT_MAX;
  for (int64_t w = gid; w < d.n_words; w += gsz) {
    const unsigned bit = w == sw ? sbit : 0u;
    d.visited[w] = closed0[w] | bit;
    d.fbits[0][w] = bit;
    d.fbits[1][w] = 0u;
  }
}
__global__ void bfs_reset_kernel(int32_t* dist, int64_t V, dobfs_args d, const unsigned* closed0) {
  bfs_reset_body(dist, V, d, closed0, -1);
}

// closed0: bit v set <=> v has no in-edges (it can never be discovered bottom-up).  Once per graph.
__global__ void bfs_closed0_kernel(const int32_t* t_ro, int32_t V, unsigned* out, int n_words) {
  const int lane = dev::lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t ch = wave; ch < n_words / 2; ch += n_waves) {
    const int64_t v = ch * 64 + lane;
    const bool no_in = v < V ? (t_ro[v + 1] == t_ro[v]) : true;
    const unsigned long long m = dev::ballot(no_in);
    if (lane == 0) {
      out[2 * ch] = (unsigned)m;
      out[2 * ch + 1] = (unsigned)(m >> 32);
    }
  }
}

// heads[v] = {first, second in-neighbour of v} (-1: none); bit 30 of the second (BU_MORE) says that v has more than two
// in-edges, so a bottom-up round needs no row offsets at all (V < 2^29 on this path: direction optimisation asks for
// E >= 4 V and E < 2^31).  Once per graph (bottom-up probes, dobfs_args::heads).
__global__ void bfs_heads_kernel(const int32_t* t_ro, const int32_t* t_ci, int32_t V, int2* heads) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    const int b = t_ro[v], e = t_ro[v + 1];
    heads[v] = make_int2(b < e ? t_ci[b] : -1, b + 1 < e ? (t_ci[b + 1] | (b + 2 < e ? BU_MORE : 0)) : -1);
  }
}

// source_level: 1 = the launch that follows (bfs_source_kernel) expands the source itself when source_level_applies();
// this kernel then leaves the control block as the head of level 0 would have (level = 0, its frontier accounted for)
__device__ __forceinline__ bool source_level_applies(int deg, const dobfs_args& d) {
  // small sources: the LDS-resident tiny levels of the first head kernel take them (and the levels behind them);
  // a source with more out-edges than 1/alpha of the graph would be expanded bottom-up by the direction rule
  return deg > TINY_EDGES && !(d.enabled && (long long)deg > d.n_edges / (long long)(d.alpha > 0 ? d.alpha : 1));
}

// The seed of a search (prepare_frontier of the reference's enactor): one workgroup of TILE threads.  labels: also write the
// source's label and bitmap bits (false: the fill of bfs_reset_seed_kernel already did).
__device__ __forceinline__ void bfs_seed_body(const pipe_args& a, int32_t* dist, unsigned* visited, int src, const dobfs_args& d,
                                              int source_level, bool labels) {
  const int tid = threadIdx.x;
  int32_t* f0 = a.frontier[0];
  f0[tid] = (tid == 0) ? src : -1;
  if (tid == 0) {
    ctrl_t* c = a.ctrl;
    const int deg = a.ro[src + 1] - a.ro[src];
    a.tile_sums[0] = deg;
    a.tile_chunks[0] = (deg + CHUNK - 1) / CHUNK;
    a.tile_count[0] = 1;
    c->level = -1;
    c->done = 0;
    c->n_tiles[0] = 1;
    c->n_items[0] = 1;
    c->n_tiles[1] = 0;
    c->n_items[1] = 0;
    c->total_chunks = 0;
    c->map_chunks = 0;
    c->map_level = -2;
    c->edges_visited = 0;
    c->vertices_visited = 0;
    c->g_edges_visited = 0;
    c->spare[0] = 0;
    c->bin_want = 0;
    c->mode = 0;
    c->frontier_bitmap = 0;
    c->convert = 0;
    c->q_edges[0] = deg;
    c->q_edges[1] = 0;
    c->bu_open = 0;
    c->bu_probes = 0;
    c->bu_R = 0;
    c->bu_T = 0;
    c->t_start = (long long)wall_clock64();
    a.mailbox[0] = 0;
    a.mailbox[1] = 0;
    a.mailbox[2] = 0;
    if (source_level && source_level_applies(deg, d)) {
      // what the head of level 0 leaves behind (plan_body / bfs_decide_body on a one-vertex frontier)
      c->level = 0;
      c->edges_visited = deg;
      c->vertices_visited = 1;
      c->n_tiles[0] = 0;
      c->total_chunks = (deg + CHUNK - 1) / CHUNK;
      a.mailbox[1] = 0;
      a.mailbox[2] = 1;
      // source_level == 2: bfs_source_kernel also writes the chunk map and the counters of level 1 (map_chunks, n_items[1] and
      // q_edges[1] start at 0 above), as the sweep of a binned level does for the level behind it
      if (source_level == 2) c->map_level = 1;
    }
    if (!labels) return;
    dist[src] = 0;
    if (d.enabled) {
      const unsigned bit = 1u << (src & 31);
      d.visited[src >> 5] |= bit;
      d.fbits[0][src >> 5] = bit;
    } else if (visited) {
      visited[src >> 5] = 1u << (src & 31);
    }
  }
}

__global__ void bfs_init_kernel(pipe_args a, int32_t* dist, unsigned* visited, int src, dobfs_args d, int source_level) {
  bfs_seed_body(a, dist, visited, src, d, source_level, true);
}
// problem.reset() and the seed of a direction-optimising search in ONE launch (a launch costs ~4 us whatever it does): the
// fill writes the source's label and bits itself, workgroup 0 sets the control block up.  The search's own clock
// (ctrl_t::t_start) starts with this kernel, so the reported enact() time now INCLUDES the reset.  <<<any, TILE>>>
__global__ void bfs_reset_seed_kernel(int32_t* dist, int64_t V, dobfs_args d, const unsigned* closed0, pipe_args a, int src,
                                      int source_level) {
  if (blockIdx.x == 0) bfs_seed_body(a, dist, nullptr, src, d, source_level, false);
  bfs_reset_body(dist, V, d, closed0, src);
}

// The same for a forward-only search that keeps a visited bitmap: labels = INT_MAX (16-byte stores), visited = 0, the source's
// label and bit written by the fill, the seed in workgroup 0 -- one launch where fill_i32 + hipMemsetAsync + bfs_init_kernel
// were three (9 + 4 + 4 us on the LJ stand-in).  dist: 16-byte aligned (the host checks); n_words: a multiple of 4.
// <<<any, TILE>>>
__global__ void bfs_fwd_reset_seed_kernel(int32_t* dist, int64_t V, unsigned* visited, int n_words, pipe_args a, int src,
                                          dobfs_args d, int source_level) {
  if (blockIdx.x == 0) bfs_seed_body(a, dist, nullptr, src, d, source_level, false);
  const int64_t gsz = (int64_t)gridDim.x * blockDim.x, gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = V >> 2, s4 = (int64_t)src >> 2;
  int4* d4 = reinterpret_cast<int4*>(dist);
  for (int64_t i = gid; i < n4; i += gsz) {
    int4 v = make_int4(INT_MAX, INT_MAX, INT_MAX, INT_MAX);
    if (i == s4) {
      const int k = src & 3;
      if (k == 0) v.x = 0; else if (k == 1) v.y = 0; else if (k == 2) v.z = 0; else v.w = 0;
    }
    d4[i] = v;
  }
  for (int64_t i = (n4 << 2) + gid; i < V; i += gsz) dist[i] = i == (int64_t)src ? 0 : INT_MAX;
  uint4* b4 = reinterpret_cast<uint4*>(visited);
  const int64_t w4 = (int64_t)(src >> 5) >> 2;
  for (int64_t i = gid; i < (int64_t)(n_words >> 2); i += gsz) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (i == w4) {
      const unsigned bit = 1u << (src & 31);
      const int k = (src >> 5) & 3;
      if (k == 0) v.x = bit; else if (k == 1) v.y = bit; else if (k == 2) v.z = bit; else v.w = bit;
    }
    b4[i] = v;
  }
}

// LEVEL 0 WITHOUT A LAUNCH PAIR.  Every search starts from one vertex; when that vertex is a hub (the benchmark sources
// are: 125 k out-edges on the LJ stand-in) level 0 used to cost a head kernel (12 us: bookkeeping and a chunk map for ONE
// tile) plus a level kernel (27 us: the general advance body, staging a 256-slot tile of which one slot is used).  Here
// the seed kernel does the bookkeeping and this kernel expands the source's row directly: lanes on consecutive edges, the
// policy's claim (no probe first: nothing but the source is visited), discoveries compacted into tiles exactly as
// advance_block emits them.  The first level group then starts at level 1.  <<<ceil(deg / 2048) capped, 256>>>
struct source_smem {
  int out[TILE + CHUNK];
  int wave[ADV_BLOCK / 64 + 1];
  int cnt;
  int res[3];
  emit_smem emit;
  int cpre[MAX_EMIT + 1];  // write_map: chunks before tile j of the emission
  int map_n, map_base;
};
// write_map (forward runs with binned levels): the chunk map entries and the counters of the tiles just emitted -- sm.emit.tix /
// sm.emit.sum of k full tiles (k > 0), or the one short tile sm.res[2] / sm.wave (k == 0, n vertices) -- are appended by
// the producers (one reservation atomic on ctrl.map_chunks per call), so the head of level 1 has nothing to walk (it took
// 18 us to plan the 349 tiles / 15 k chunks behind the LJ stand-in's source, 35 / 45 us on the kron / twitter stand-ins).
// Block-wide call right behind the emission.
static_assert(MAX_EMIT + 1 <= 64, "one lane per tile of an emission");
__device__ __forceinline__ void source_append_map(const pipe_args& a, ctrl_t* c, source_smem& sm, int k, int n) {
  const int tid = threadIdx.x, lane = dev::lane_id();
  const int nt = k > 0 ? k : 1;
  if (tid < 64) {
    int tot = 0;
    if (lane < nt) {
      if (k > 0) {
#pragma unroll
        for (int i = 0; i < ADV_BLOCK / 64; ++i) tot += sm.emit.sum[lane][i];
      } else {
#pragma unroll
        for (int i = 0; i < ADV_BLOCK / 64; ++i) tot += sm.wave[i];
      }
    }
    const int ch = (tot + CHUNK - 1) / CHUNK;
    const int inc = dev::wave_inclusive_sum(ch);
    long long es = (long long)tot;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) es += __shfl_xor(es, o, 64);
    if (lane <= MAX_EMIT) sm.cpre[lane] = inc - ch;
    if (lane == 63) {
      sm.map_n = inc;
      sm.map_base = inc > 0 ? atomicAdd(&c->map_chunks, inc) : 0;
    }
    if (lane == 0) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&c->q_edges[1]), (unsigned long long)es);
      atomicAdd(&c->n_items[1], k > 0 ? k * TILE : n);
    }
  }
  __syncthreads();
  int2* map = reinterpret_cast<int2*>(a.chunk_tile) + sm.map_base;
  const int nc = sm.map_n;
  for (int ci = tid; ci < nc; ci += ADV_BLOCK) {
    int t = 0;  // largest t < nt with cpre[t] <= ci (tiles without chunks are skipped over)
#pragma unroll
    for (int step = 8; step >= 1; step >>= 1)
      if (t + step < nt && sm.cpre[t + step] <= ci) t += step;
    map[ci] = make_int2(k > 0 ? sm.emit.tix[t] : sm.res[2], ci - sm.cpre[t]);
  }
  __syncthreads();
}
__global__ __launch_bounds__(ADV_BLOCK) void bfs_source_kernel(pipe_args a, dobfs_args d, bfs_policy pol, int src, int write_map) {
  __shared__ source_smem sm;
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  const int rs = a.ro[src];
  const int deg = a.ro[src + 1] - rs;
  if (!source_level_applies(deg, d)) return;
  ctrl_t* c = a.ctrl;
  pol.ctrl = c;
  pol.set_level(0);
  if (d.enabled) {
    // level 0's share of the bitmap upkeep of a direction-optimising run: the frontier bitmap level 1 writes into
    uint4* z = reinterpret_cast<uint4*>(pick3(d.fbits, 2));
    for (int i = blockIdx.x * ADV_BLOCK + tid; i < d.n_words / 4; i += gridDim.x * ADV_BLOCK) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (tid == 0) { sm.cnt = 0; sm.res[0] = 0; sm.res[1] = 0; }
  __syncthreads();
  for (int a0 = (int)blockIdx.x * CHUNK; a0 < deg; a0 += (int)gridDim.x * CHUNK) {
    int n_k[ADV_ITEMS], r_k[ADV_ITEMS];
    bool ok_k[ADV_ITEMS];
#pragma unroll
    for (int k = 0; k < ADV_ITEMS; ++k) {
      const int i = a0 + k * ADV_BLOCK + tid;
      ok_k[k] = i < deg;
      n_k[k] = a.ci[rs + (ok_k[k] ? i : 0)];
    }
    // Runs that keep bitmaps claim ON THE BITMAP: the bit every discovery has to set in the next frontier (or, forward-only
    // runs, in `visited`) is set with a returning atomic and decides -- nothing but the source is visited at level 0 -- and the
    // label is a plain store: one atomic per edge where the label claim + the bit took two (571 k out-edges of the kron
    // stand-in's source: the level is bound by the rate of those atomics).
    const bool by_bit = pol.bm_visited != nullptr;  // uniform
#pragma unroll
    for (int k = 0; k < ADV_ITEMS; ++k) {
      r_k[k] = 0;
      if (ok_k[k]) {
        if (!by_bit) r_k[k] = pol.claim(n_k[k], 0);
        else if (n_k[k] == src) r_k[k] = -1;  // a self loop: the source is visited, and no bit of the next frontier
        else r_k[k] = (int)__hip_atomic_fetch_or(&pol.bm_next[n_k[k] >> 5], 1u << (n_k[k] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#pragma unroll
    for (int k = 0; k < ADV_ITEMS; ++k) {
      bool keep;
      if (by_bit) keep = ok_k[k] && (((unsigned)r_k[k] >> (n_k[k] & 31)) & 1u) == 0u;
      else keep = ok_k[k] && pol.code(r_k[k], 0, n_k[k], 0) == 1;
      const unsigned long long m = dev::ballot(keep);
      if (m) {
        int at = 0;
        if (lane == 0) at = atomicAdd(&sm.cnt, __popcll(m));
        at = dev::wave_bcast0(at);
        if (keep) {
          sm.out[at + dev::mask_rank(m)] = n_k[k];
          if (by_bit) pol.dist[n_k[k]] = pol.next_depth;
          else pol.on_accept(n_k[k]);
        }
      }
    }
    __syncthreads();
    int cnt = sm.cnt;
    if (cnt >= TILE) {
      const int k = cnt / TILE;
      emit_full_tiles(a, c, 1, sm.out, cnt - k * TILE, k, sm.emit, sm.res);
      if (write_map) source_append_map(a, c, sm, k, 0);
      cnt -= k * TILE;
    }
    if (tid == 0) sm.cnt = cnt;
    __syncthreads();
  }
  const int rem = sm.cnt;
  if (rem > 0) {
    emit_tile(a, c, 1, sm.out, 0, rem, sm.wave, sm.res);
    __syncthreads();
    if (write_map) source_append_map(a, c, sm, 0, rem);
  }
  __syncthreads();
  release_tiles(a, sm.res);
}

// Level bookkeeping + direction choice (one workgroup of PLAN_BLOCK threads).
// The size of the frontier entering this level is reduced here from per-tile counts (after a
// top-down level) or per-workgroup partials (after a bottom-up level), so producers need no
// counter atomics.
// s_red: 4 LDS words, zeroed and synchronised on entry.
// s_dec (3 LDS ints) receives {done, direction of this level, level} for every thread.
// only_finish (the group is this head alone, plan_in::only_finish): a search that is over is finished as usual, one that is not
// is left exactly as it is (s_dec[1] = -1: nothing follows) for the head of the next group.
// PART (partitioned searches, part_args): the sums above are this rank's share -- they feed its own counters -- while the end of
// the search and the direction are decided from the ALL-REDUCED frontier statistics, so every rank takes the same branch.
template <bool PART = false>
__device__ __forceinline__ void bfs_decide_body(const pipe_args& a, const dobfs_args& d, unsigned long long* s_red,
                                                const ctrl_head& h, int* s_dec, int only_finish = 0,
                                                const part_args* x = nullptr) {
  ctrl_t* c = a.ctrl;
  const int tid = threadIdx.x;
  const int done = h.done;
  const int level = h.level + 1;
  const int p = level & 1;
  const int prev_bottom_up = h.mode;
  const int nt = h.nt(p);
  if (tid == 0) { s_dec[0] = done; s_dec[1] = prev_bottom_up; s_dec[2] = level; }
  if (done) { __syncthreads(); return; }
  long long n = 0, m = 0, op = 0, pr = 0;
  if (prev_bottom_up) {
    // a bottom-up level leaves one record per workgroup (its tiles span a sparse static range)
    for (int i = tid; i < d.bu_grid; i += PLAN_BLOCK) {
      n += d.bu_part[4 * i] & ((1ll << 40) - 1);
      m += d.bu_part[4 * i + 1];
      op += d.bu_part[4 * i + 2];
      pr += d.bu_part[4 * i + 3];
    }
  } else {
    for (int i = tid; i < nt; i += PLAN_BLOCK) {
      n += a.tile_count[i];
      m += a.tile_sums[i];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_xor(n, o, 64);
    m += __shfl_xor(m, o, 64);
    op += __shfl_xor(op, o, 64);
    pr += __shfl_xor(pr, o, 64);
  }
  if (dev::lane_id() == 0) {
    atomicAdd(&s_red[0], (unsigned long long)n);
    atomicAdd(&s_red[1], (unsigned long long)m);
    atomicAdd(&s_red[2], (unsigned long long)op);
    atomicAdd(&s_red[3], (unsigned long long)pr);
  }
  __syncthreads();
  if (tid == 0) {
    const long long n_f = (long long)s_red[0], m_f = (long long)s_red[1];
    // what the rules below look at: the whole frontier (PART: all ranks' shares)
    const long long gn = PART ? x->stats_global[0] : n_f, gm = PART ? x->stats_global[1] : m_f;
    if (only_finish && gn != 0) {
      s_dec[1] = -1;
    } else {
    c->bu_open += (long long)s_red[2];
    c->bu_probes += (long long)s_red[3];
    if (gn == 0) {
      c->done = 1;
      c->level = level;
      s_dec[0] = 1;
      publish_done(a, c, level);
    } else {
      int mode = prev_bottom_up;
      const long long e_all = PART ? x->e_global : (long long)d.n_edges;
      const long long m_u = e_all - (PART ? c->g_edges_visited : h.edges_visited);  // edges of still-unexpanded vertices
      if (PART) c->g_edges_visited += gm;
      if (mode == 0) {
        if (gm > m_u / d.alpha && gn > 256) mode = 1;
      } else {
        // back to top-down: few frontier vertices (Beamer), or so few frontier out-edges that
        // expanding them beats another sweep over every open vertex's in-edges
        if (gn < (long long)a.V / d.beta) mode = 0;
        if (d.back_div > 0 && gm < e_all / d.back_div) mode = 0;
      }
      if (!prev_bottom_up) c->bu_R = 0;  // the queue of this level is a dense run of tiles
      s_dec[1] = mode;
      c->mode = mode;
      c->level = level;
      c->edges_visited += m_f;
      c->vertices_visited += n_f;
      c->n_items[p] = (int)n_f;
      c->q_edges[p] = m_f;
      c->n_tiles[p ^ 1] = 0;
      a.mailbox[1] = level;
      a.mailbox[2] = (int)n_f;
    }
    }
  }
  __syncthreads();
}

// Head of a level, ONE launch of one workgroup: as many tiny levels as there are
// (tiny_levels_body), then level bookkeeping + direction choice, then the chunk map of a
// top-down level.  (A trivial kernel costs ~4 us on this part: this used to be three.)  <<<1, 1024>>>
__global__ __launch_bounds__(PLAN_BLOCK) void bfs_head_kernel(pipe_args a, dobfs_args d, bfs_policy pol,
                                                              int allow_tiny, int seq, bin_args bn) {
  __shared__ tiny_smem<bfs_policy> tsm;
  __shared__ unsigned long long s_red[4];
  __shared__ int s_wave[PLAN_BLOCK / 64 + 1];
  static_assert(TINY_THREADS == PLAN_BLOCK, "head kernel runs both bodies");
  __shared__ int s_dec[3];
  ctrl_t* c = a.ctrl;
  const ctrl_head h0 = load_ctrl_head(c);
  if (threadIdx.x < 4) s_red[threadIdx.x] = 0ull;
  // host pacing: group `seq` has started (not after `done`: such a launch may still be in flight
  // when the host has already begun the next search)
  if (threadIdx.x == 0 && !h0.done) a.mailbox[3] = seq;
  __syncthreads();
  const int t = allow_tiny ? tiny_levels_body(a, pol, d.enabled, (long long)d.n_edges, tsm, h0) : 0;
  if (t == 1) return;
  const ctrl_head h = t == 2 ? load_ctrl_head(c) : h0;  // tiny levels ran and handed back: fresh state
  plan_in in;
  in.done = h.done;
  in.mode = 0;
  in.R = 0;
  in.T = h.bu_T;
  if (!d.enabled) {
    in.level = h.level + 1;
    in.nt = h.nt(in.level & 1);
    in.bin_min = bn.min_edges;  // > 0: fat levels run binned (mode 2), see grx_bin.hpp
    in.bin_max_degree = bn.max_degree;
    in.mid_v = bn.mid_v;
    in.mid_e = bn.mid_e;
    in.mid_tile_e = bn.mid_tile_e;
    in.bin_early_div = bn.bin_early_div;
    in.bin_fill = bn.fill;
    in.bin_queue = bn.queue;
    in.bin_nb = bn.nb;
    in.bin_pad = BIN_PAD;
    in.bin_allowed = bn.allowed;
    in.bin_forced = bn.no_level;
    in.only_finish = bn.only_finish;
    in.seq = seq;
    plan_body<PLAN_BLOCK>(a, c, 0, s_wave, &s_red[0], in);
    return;
  }
  bfs_decide_body(a, d, s_red, h, s_dec, bn.only_finish);
  if (s_dec[0] || s_dec[1] != 0) return;
  if (threadIdx.x < 2) s_red[threadIdx.x] = 0ull;
  __syncthreads();
  in.level = s_dec[2];
  in.nt = h.nt(in.level & 1);
  in.R = h.mode ? h.bu_R : 0;  // the previous level ran bottom-up: its tiles sit in static ranges
  plan_body<PLAN_BLOCK>(a, c, 1, s_wave, &s_red[0], in);
}

// One level, ONE launch: top-down (advance + fused compaction) or bottom-up, as the
// head kernel decided.  Direction-optimising runs: every workgroup also clears its share of
// the frontier bitmap the NEXT level will write into.
// only_mode (partitioned searches): -1 any; 0 | 1: this launch runs the level only when it is top-down | bottom-up -- the exchange
// of a partitioned level sits BEHIND a top-down body and IN FRONT OF a bottom-up one, so such a group carries the kernel twice.
template <int BATCH, bool BU2, bool DBG, class Pol>
__device__ __forceinline__ void bfs_level_body(const pipe_args& a, const dobfs_args& d, Pol& pol, int only_mode) {
  // a launch runs ONE of the two bodies: their LDS is overlaid (24 KB instead of 38: 6 workgroups per CU)
  using td_smem = advance_smem<Pol>;
  using bu_smem = typename std::conditional<BU2, bottomup2_smem<BATCH>, bottomup_smem<BATCH, true>>::type;
  constexpr size_t LDS_BYTES = sizeof(td_smem) > sizeof(bu_smem) ? sizeof(td_smem) : sizeof(bu_smem);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  td_smem& sm = *reinterpret_cast<td_smem*>(lds_raw);
  bu_smem& bsm = *reinterpret_cast<bu_smem*>(lds_raw);
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);  // one batch of loads
  if (h.done) return;
  const int level = h.level;
  const int mode_now = h.mode;
  if (only_mode >= 0 && mode_now != only_mode) return;

  if (d.enabled) {
    uint4* z = reinterpret_cast<uint4*>(pick3(d.fbits, (level + 2) % 3));
    const uint4* fc = reinterpret_cast<const uint4*>(pick3(d.fbits, level % 3));
    uint4* vis = reinterpret_cast<uint4*>(d.visited);
    const bool fold = mode_now == 0;  // top-down: the frontier being expanded joins `visited` here
    const int n4 = d.n_words / 4;    // n_words is a multiple of 4 (bitmaps are padded)
    for (int i = blockIdx.x * ADV_BLOCK + threadIdx.x; i < n4; i += gridDim.x * ADV_BLOCK) {
      z[i] = make_uint4(0u, 0u, 0u, 0u);
      if (fold) {
        const uint4 f = fc[i];
        if (f.x | f.y | f.z | f.w) {
          uint4 v = vis[i];
          v.x |= f.x; v.y |= f.y; v.z |= f.z; v.w |= f.w;
          vis[i] = v;
        }
      }
    }
  }
  if (mode_now == 0) {
    // (a frontier left by a bottom-up level sits in that launch's static tile ranges: thousands of chunks of a few vertices each,
    // cheapest at one per workgroup -- 9 against 12 us on the LJ stand-in's fourth level)
    const int n_act = c->bu_R > 0 ? (int)gridDim.x : thin_workgroups(h.total_chunks, (int)gridDim.x, d.thin_div, d.thin_min);
    if ((int)blockIdx.x >= n_act) return;
    pol.ctrl = c;
    pol.set_level(level);
    advance_block<Pol, false>(a, c, pol, sm, level & 1, blockIdx.x, n_act, h.total_chunks, a.chunk_tile);
  } else {
    if constexpr (BU2) bfs_bottomup2_block<BATCH, DBG>(a, d, c, bsm);
    else bfs_bottomup_block<BATCH, true>(a, d, c, bsm);
  }
}
template <int BATCH, bool BU2 = false, bool DBG = false>
__global__ __launch_bounds__(ADV_BLOCK) void bfs_level_kernel(pipe_args a, dobfs_args d, bfs_policy pol) {
  bfs_level_body<BATCH, BU2, DBG>(a, d, pol, -1);
}
template <int BATCH, bool BU2 = false>
__global__ __launch_bounds__(ADV_BLOCK) void bfs_level_part_kernel(pipe_args a, dobfs_args d, bfs_policy_part pol, int only_mode) {
  bfs_level_body<BATCH, BU2, false>(a, d, pol, only_mode);
}

// Forward-only runs with binned fat levels: one level is this kernel (the claim-per-edge advance, or
// the SCATTER phase of a binned level) followed by bfs_claim_kernel (a no-op unless the level is
// binned).  Separate from bfs_level_kernel so that the 38 KB of LDS the scatter phase sorts in do
// not cost the direction-optimising path its resident workgroups.
__global__ __launch_bounds__(ADV_BLOCK) void bfs_level_bin_kernel(pipe_args a, bin_args bn, bfs_policy pol) {
  using td_smem = mid_smem<bfs_policy>;
  constexpr size_t LDS_BYTES = sizeof(td_smem);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);
  if (h.done) return;
  if (h.mode == 2) {
    return;  // a binned level: its kernels are bfs_scatter2_kernel + bfs_sweep2_kernel, the next launches of the group
  } else if (h.mode == 3) {
    pol.ctrl = c;
    mid_levels_run(a, c, pol, *reinterpret_cast<td_smem*>(lds_raw), h, bn.xcc_mask);
  } else {
    const int n_act = thin_workgroups(h.total_chunks, (int)gridDim.x, bn.thin_div, bn.thin_min);
    if ((int)blockIdx.x >= n_act) return;
    pol.ctrl = c;
    pol.set_level(h.level);
    advance_block<bfs_policy, false>(a, c, pol, reinterpret_cast<td_smem*>(lds_raw)->adv, h.level & 1, blockIdx.x,
                                     n_act, h.total_chunks, a.chunk_tile);
  }
}

// The scatter phase of a binned level, second version (grx_bin.hpp): 1024 threads, two workgroups = 32 waves per CU.
template <bool DBG, bool E16, bool UNI = false>
__global__ __launch_bounds__(SC2_BLOCK, 8) void bfs_scatter2_kernel(pipe_args a, bin_args bn) {
  __shared__ __attribute__((aligned(16))) bin_scatter2_smem sm;
  const level_head h = load_level_head(a.ctrl);
  if (h.done || h.mode != 2) return;
  if (DBG && h.level != bn.debug_level) {
    bin_scatter2_block<false, E16, false, bin_scatter2_smem, UNI>(a, bn, sm, h.level & 1, h.total_chunks, a.chunk_tile);
    return;
  }
  bin_scatter2_block<DBG, E16, false, bin_scatter2_smem, UNI>(a, bn, sm, h.level & 1, h.total_chunks, a.chunk_tile);
}

// The claim phase as a sweep (grx_bin.hpp): NT = 1024 threads, <= 128 VGPRs, a 16128-entry list (82 KB of LDS): one workgroup
// per CU, one part per bin unless it is fat, ONE emission per item.
template <int NT, int LE, int WAVES_PER_SIMD, bool DBG, bool E16>
__global__ __launch_bounds__(NT, WAVES_PER_SIMD) void bfs_sweep2_kernel(pipe_args a, bin_args bn) {
  __shared__ __attribute__((aligned(16))) bin_sweep2_smem<NT, LE> sm;
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);
  if (h.done || h.mode != 2) return;
  if (DBG && h.level != bn.debug_level) {
    bin_sweep2_block<NT, LE, false, E16>(a, bn, c, h.level + 1, sm, h.level & 1);
    return;
  }
  bin_sweep2_block<NT, LE, DBG, E16>(a, bn, c, h.level + 1, sm, h.level & 1);
}
constexpr int SW3_BLOCK = 1024, SW3_LIST = 63 * TILE;

// ===============================================================================================================
// PARTITIONED searches (round 6): the kernels a partition adds around the engine's bodies.  One level group of rank r:
//   forward:   head -> level (claim-per-edge) | scatter + sweep  -> [exchange of `send`] -> post -> stats -> [all-reduce]
//   direction-optimising: head -> prep -> level(top-down only) -> [exchange] -> level(bottom-up only) -> post -> stats -> [all-reduce]
// The bracketed collectives are the host's (RCCL inside the library, or torch.distributed); nothing here depends on them
// except through part_args::recv / stats_global.  See bfs_policy_part and part_args (grx_bfs_kernels.hpp).
// ===============================================================================================================

// Head of a partitioned level group: as bfs_head_kernel without the tiny levels (a level ends with an exchange), the end of
// the search and the direction taken from the all-reduced statistics.  <<<1, 1024>>>
__global__ __launch_bounds__(PLAN_BLOCK) void bfs_head_part_kernel(pipe_args a, dobfs_args d, int seq, bin_args bn, part_args x) {
  __shared__ unsigned long long s_red[4];
  __shared__ int s_wave[PLAN_BLOCK / 64 + 1];
  __shared__ int s_dec[3];
  ctrl_t* c = a.ctrl;
  const ctrl_head h = load_ctrl_head(c);
  if (threadIdx.x < 4) s_red[threadIdx.x] = 0ull;
  if (threadIdx.x == 0 && !h.done) a.mailbox[3] = seq;
  __syncthreads();
  plan_in in;
  in.done = h.done;
  in.mode = 0;
  in.R = 0;
  in.T = h.bu_T;
  in.part_P = x.P;
  if (!d.enabled) {
    in.level = h.level + 1;
    in.nt = h.nt(in.level & 1);
    in.g_n = x.stats_global[0];
    in.bin_min = bn.min_edges;
    in.bin_max_degree = bn.max_degree;
    in.bin_early_div = bn.bin_early_div;
    in.bin_fill = bn.fill;
    in.bin_queue = bn.queue;
    in.bin_nb = bn.nb;
    in.bin_pad = BIN_PAD;
    in.bin_allowed = bn.allowed;
    in.seq = seq;
    plan_body<PLAN_BLOCK>(a, c, 0, s_wave, &s_red[0], in);
    return;
  }
  bfs_decide_body<true>(a, d, s_red, h, s_dec, 0, &x);
  if (s_dec[0] || s_dec[1] != 0) return;
  if (threadIdx.x < 2) s_red[threadIdx.x] = 0ull;
  __syncthreads();
  in.level = s_dec[2];
  in.nt = h.nt(in.level & 1);
  in.R = h.mode ? h.bu_R : 0;
  plan_body<PLAN_BLOCK>(a, c, 1, s_wave, &s_red[0], in);
}

// Forward partitioned run, thin level: the claim-per-edge advance with the partition's claim (a binned level: no-op).
__global__ __launch_bounds__(ADV_BLOCK) void bfs_level_bin_part_kernel(pipe_args a, bin_args bn, bfs_policy_part pol) {
  __shared__ advance_smem<bfs_policy_part> sm;
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);
  if (h.done || h.mode != 0) return;
  const int n_act = thin_workgroups(h.total_chunks, (int)gridDim.x, bn.thin_div, bn.thin_min);
  if ((int)blockIdx.x >= n_act) return;
  pol.ctrl = c;
  pol.set_level(h.level);
  advance_block<bfs_policy_part, false>(a, c, pol, sm, h.level & 1, blockIdx.x, n_act, h.total_chunks, a.chunk_tile);
}

// ... fat level: the engine's scatter (bfs_scatter2_kernel, unchanged: bins span the whole vertex range) and its sweep with the
// partition's rule for words of other ranks' vertices (bin_sweep2_block<.., SLICED>).
template <int NT, int LE, int WAVES_PER_SIMD, bool E16>
__global__ __launch_bounds__(NT, WAVES_PER_SIMD) void bfs_sweep2_part_kernel(pipe_args a, bin_args bn) {
  __shared__ __attribute__((aligned(16))) bin_sweep2_smem<NT, LE> sm;
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);
  if (h.done || h.mode != 2) return;
  bin_sweep2_block<NT, LE, false, E16, true>(a, bn, c, h.level + 1, sm, h.level & 1);
}

// problem.reset() + seed of a partitioned search, one launch: labels of the OWNED range only (the label pointer may be the
// base of a sharded array minus lo), bitmaps of the whole range, an empty outgoing bitmap; the frontier is {source} on the
// owner and empty elsewhere; the first statistics record.  <<<any, TILE>>>
__global__ void bfs_part_reset_seed_kernel(int32_t* dist, pipe_args a, dobfs_args d, part_args x, unsigned* visited, int n_words,
                                           const unsigned* closed0, int src) {
  const bool own = src >= x.lo && src < x.hi;
  if (blockIdx.x == 0) {
    bfs_seed_body(a, dist, nullptr, src, d, 0, false);
    if (threadIdx.x == 0) {
      ctrl_t* c = a.ctrl;
      const int deg = own ? a.ro[src + 1] - a.ro[src] : 0;
      if (!own) {
        a.frontier[0][0] = -1;
        a.tile_sums[0] = 0;
        a.tile_chunks[0] = 0;
        a.tile_count[0] = 0;
        c->n_tiles[0] = 0;
        c->n_items[0] = 0;
        c->q_edges[0] = 0;
      }
      x.stats_local[0] = own ? 1 : 0;
      x.stats_local[1] = deg;
      x.stats_local[2] = 0;
      x.stats_local[3] = 0;
    }
  }
  const int64_t gsz = (int64_t)gridDim.x * blockDim.x, gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = (int64_t)x.lo + gid; i < (int64_t)x.hi; i += gsz) dist[i] = i == (int64_t)src ? 0 : INT_MAX;
  const int64_t sw = src >> 5;
  const unsigned sbit = 1u << (src & 31);
  for (int64_t w = gid; w < n_words; w += gsz) {
    const unsigned bit = w == sw ? sbit : 0u;  // (every rank marks the source: nobody reports it to its owner)
    if (d.enabled) {
      visited[w] = closed0[w] | bit;
      d.fbits[0][w] = bit;
      d.fbits[1][w] = 0u;
    } else {
      visited[w] = bit;
    }
  }
  const int64_t all = (int64_t)x.P * x.slice_words;
  for (int64_t w = gid; w < all; w += gsz) {
    x.send[w] = 0u;
    if (d.enabled) x.sent[w] = w == sw ? sbit : 0u;
  }
}

// Direction-optimising partitioned run, in front of the exchange of a BOTTOM-UP level: this rank's slice of the frontier
// bitmap goes to every peer (the all-to-all degenerates to an all-gather: afterwards `recv` is the whole-graph frontier).
__global__ void bfs_part_prep_kernel(pipe_args a, dobfs_args d, part_args x) {
  const level_head h = load_level_head(a.ctrl);
  if (h.done || h.mode != 1) return;
  const unsigned* f = pick3(d.fbits, h.level % 3) + (x.lo >> 5);
  const int n_w = (x.hi - x.lo + 31) >> 5;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < x.slice_words; i += gridDim.x * blockDim.x) {
    const unsigned v = i < n_w ? f[i] : 0u;
    for (int j = 0; j < x.P; ++j) x.send[(size_t)j * x.slice_words + i] = v;
  }
}

// Behind the exchange: what the peers reported in this rank's slice -- OR of the P - 1 received slices, minus what is
// known (`visited`; direction-optimising runs: + the frontier bitmap being built) -- is claimed (exactly one thread looks at a
// word), labelled and emitted in ascending vertex order as tiles of the next frontier; behind a BINNED level also with the
// chunk-map entries and counters the sweep's own emission writes (sweep2_emit), so that the next head has nothing to walk.
// Every launch also empties the outgoing bitmap for the next level.  <<<min(segments, 2 per CU), 1024>>>
constexpr int PP_BLOCK = 1024, PP_SEG_WORDS = PP_BLOCK / 4;
struct part_post_smem {
  static constexpr int LIST = 34 * TILE;  // a segment's 8192 ids + what is waiting (a short tile at most after an emission)
  static constexpr int MAX_TILES = LIST / TILE + 1;
  int list[LIST];
  unsigned words[PP_SEG_WORDS];
  int wave[PP_BLOCK / 64 + 1];
  int sum[MAX_TILES][4];
  int ttot[64];
  int cpre[64];
  int tile_base;
  int chunk_base;
  int n_chunks;
};
static_assert(part_post_smem::LIST >= PP_SEG_WORDS * 32 + 2 * TILE, "a segment fits behind a short tile");
__global__ __launch_bounds__(PP_BLOCK) void bfs_part_post_kernel(pipe_args a, dobfs_args d, part_args x, unsigned* visited) {
  __shared__ __attribute__((aligned(16))) part_post_smem sm;
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);
  if (h.done) return;
  const int tid = threadIdx.x;
  {
    uint4* z = reinterpret_cast<uint4*>(x.send);  // (slice_words is a multiple of 64)
    const int64_t n4 = (int64_t)x.P * x.slice_words / 4;
    for (int64_t i = (int64_t)blockIdx.x * PP_BLOCK + tid; i < n4; i += (int64_t)gridDim.x * PP_BLOCK) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (h.mode == 1) return;  // a bottom-up level: its discoveries are this rank's own
  const int depth = h.level + 1;
  const int q = depth & 1;
  const bool with_map = !d.enabled && __hip_atomic_load(&c->map_level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == depth;
  unsigned* fnext = d.enabled ? pick3(d.fbits, depth % 3) : nullptr;
  const int w_lo = x.lo >> 5;
  const int n_w = (x.hi - x.lo + 31) >> 5;
  const int n_seg = (n_w + PP_SEG_WORDS - 1) / PP_SEG_WORDS;
  int n_list = 0;  // uniform
  auto emit_list = [&](bool all) {
    const int k = all ? (n_list + TILE - 1) / TILE : n_list / TILE;
    const int n_emit = all ? n_list : k * TILE;
    sweep2_emit<PP_BLOCK>(a, c, q, sm, n_emit, d.dist, depth, with_map);
    const int rem = n_list - n_emit;
    int keep = 0;
    if (tid < rem) keep = sm.list[n_emit + tid];
    __syncthreads();
    if (tid < rem) sm.list[tid] = keep;
    n_list = rem;
    __syncthreads();
  };
  for (int seg = (int)blockIdx.x; seg < n_seg; seg += (int)gridDim.x) {
    if (tid < PP_SEG_WORDS) {
      const int w = seg * PP_SEG_WORDS + tid;
      unsigned nw = 0u;
      if (w < n_w) {
        unsigned cand = 0u;
        for (int j = 0; j < x.P; ++j)
          if (j != x.rank) cand |= x.recv[(size_t)j * x.slice_words + w];
        if (cand) {
          unsigned known = visited[w_lo + w];
          unsigned fn = 0u;
          if (fnext) {
            fn = fnext[w_lo + w];
            known |= fn;
          }
          nw = cand & ~known;
          if (nw) {
            if (fnext) fnext[w_lo + w] = fn | nw;  // (joins `visited` when the level kernel expands that frontier)
            else visited[w_lo + w] = known | nw;
          }
        }
      }
      sm.words[tid] = nw;
    }
    __syncthreads();
    unsigned byte = (sm.words[tid >> 2] >> ((tid & 3) * 8)) & 0xffu;
    int tot;
    const int ex = dev::block_exclusive_sum<PP_BLOCK>(__popc(byte), sm.wave, &tot);
    if (tot == 0) continue;  // (the scan's barriers separate this round's reads of `words` from the next round's writes)
    if (n_list + tot > part_post_smem::LIST) emit_list(false);
    int pos = n_list + ex;
    const int v_first = x.lo + ((seg * PP_SEG_WORDS + (tid >> 2)) << 5) + (tid & 3) * 8;
    while (byte) {
      sm.list[pos++] = v_first + __ffs(byte) - 1;
      byte &= byte - 1u;
    }
    n_list += tot;
    __syncthreads();
  }
  if (n_list > 0) emit_list(true);
}

// This rank's share of the frontier the next level expands -> part_args::stats_local, the input of the all-reduce that
// drives termination and direction on every rank.  <<<1, 1024>>>
__global__ __launch_bounds__(PLAN_BLOCK) void bfs_part_stats_kernel(pipe_args a, dobfs_args d, part_args x) {
  __shared__ unsigned long long s_red[2];
  ctrl_t* c = a.ctrl;
  const int tid = threadIdx.x;
  if (tid < 2) s_red[tid] = 0ull;
  __syncthreads();
  const ctrl_head h = load_ctrl_head(c);
  if (h.done) {
    if (tid < 4) x.stats_local[tid] = 0;
    return;
  }
  const int q = (h.level + 1) & 1;
  long long n = 0, m = 0;
  if (h.mode == 1) {
    for (int i = tid; i < d.bu_grid; i += PLAN_BLOCK) {
      n += d.bu_part[4 * i] & ((1ll << 40) - 1);
      m += d.bu_part[4 * i + 1];
    }
  } else if (!d.enabled && c->map_level == h.level + 1) {
    if (tid == 0) {  // the sweep and the post kernel summed them with the chunk map
      n = c->n_items[q];
      m = c->q_edges[q];
    }
  } else {
    const int nt = h.nt(q);
    for (int i = tid; i < nt; i += PLAN_BLOCK) {
      n += a.tile_count[i];
      m += a.tile_sums[i];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_xor(n, o, 64);
    m += __shfl_xor(m, o, 64);
  }
  if (dev::lane_id() == 0) {
    atomicAdd(&s_red[0], (unsigned long long)n);
    atomicAdd(&s_red[1], (unsigned long long)m);
  }
  __syncthreads();
  if (tid == 0) {
    x.stats_local[0] = (long long)s_red[0];
    x.stats_local[1] = (long long)s_red[1];
    x.stats_local[2] = 0;
    x.stats_local[3] = 0;
  }
}

}  // namespace grx

using namespace grx;

// Tuning knobs are read per call (tests and A/B tools change them between searches): ~35 getenv() calls = 3 us of host
// time per search, which sits between two back-to-back searches once no no-op groups trail the first.  grx_bfs scans the
// environment ONCE per call for any GRX_ variable; without one (the normal case) env_int returns its default at once.
extern char** environ;
static thread_local bool t_no_grx_env = false;  // valid inside grx_bfs only (env_scan_guard)
struct env_scan_guard {
  env_scan_guard() {
    bool any = false;
    for (char** e = environ; e && *e; ++e)
      if ((*e)[0] == 'G' && (*e)[1] == 'R' && (*e)[2] == 'X' && (*e)[3] == '_') { any = true; break; }
    t_no_grx_env = !any;
  }
  ~env_scan_guard() { t_no_grx_env = false; }
};
static int env_int(const char* name, int dflt) {
  if (t_no_grx_env) return dflt;
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// Per-graph static part of the binned levels (cached in the graph handle): the bins -- runs of
// granules with about equal numbers of in-edges -- their capacities, the XCD that claims each
// bin (longest-processing-time assignment by capacity), the E-entry bin array and the counters.
static grx_status_t graph_build_bins(grx_context_t ctx, grx_graph_t g) {
  std::lock_guard<std::recursive_mutex> lk(g->prep_mu);
  if (g->bin_state == 3) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs: a column index lies outside [0, V)");
  if (g->bin_state != 0) return GRX_SUCCESS;
  lazy_state state(&g->bin_state);  // an error return leaves it at 0 (retried by the next search), done(2) = not applicable
  prep_timer tm("bfs: bin table (granule counts + cut)", ctx->stream);
  if (g->V <= 0 || g->E <= 0 || ctx->n_xcd < 1) return state.done(2);
  int gshift = BIN_GSHIFT_MIN;
  while (gshift < 31 && (((long long)g->V + (1ll << gshift) - 1) >> gshift) > BIN_GRAN_MAX) ++gshift;
  if (gshift > BIN_SHIFT_MAX) return state.done(2);  // a bin's bitmap slice would not fit the claim kernel's LDS
  const int n_gran = (int)(((long long)g->V + (1ll << gshift) - 1) >> gshift);
  // widest bin: 65536 vertices when BIN_MAX such bins cover the graph -- then an offset inside a bin fits 16 bits and the
  // bins are written and streamed as 16-bit entries (second scatter + second sweep) -- else 131072 (32-bit entries)
  int shift_max = BIN_SHIFT_MAX;
  if (gshift <= 16) {
    const int mw16 = 1 << (16 - gshift);
    if ((n_gran + mw16 - 1) / mw16 <= BIN_MAX - 32 && env_int("GRX_BIN_ENTRY16", 1) != 0) shift_max = 16;
  }
  const int max_width = 1 << (shift_max - gshift);  // granules per bin
  if ((n_gran + max_width - 1) / max_width > BIN_MAX) return state.done(2);
  hipStream_t s = ctx->stream;
  dev_scratch cnt_buf;
  GRX_HIP(cnt_buf.alloc(BIN_GRAN_MAX * sizeof(int32_t)));
  int32_t* d_cnt = cnt_buf.as<int32_t>();
  GRX_HIP(hipMemsetAsync(d_cnt, 0, BIN_GRAN_MAX * sizeof(int32_t), s));
  hipLaunchKernelGGL(bin_count_kernel, dim3(ctx->num_cus * 4), dim3(256), 0, s, g->ci, (int64_t)g->E, gshift, n_gran, d_cnt);
  std::vector<int32_t> cnt(BIN_GRAN_MAX);
  GRX_HIP(hipMemcpyAsync(cnt.data(), d_cnt, BIN_GRAN_MAX * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  GRX_HIP(hipStreamSynchronize(s));
  long long total = 0;
  for (int i = 0; i < n_gran; ++i) total += cnt[(size_t)i];
  if (total != (long long)g->E)  // (state 3: reported by every call, not only the first)
    return state.done(3, fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs: a column index lies outside [0, V)"));
  // cut the granule sequence into <= BIN_MAX bins of about `target` in-edges each, no wider than max_width
  std::vector<int> first;  // first granule of each bin
  long long target = (total + 223) / 224;
  // UNIFORM bins (round 5): when 16-bit entries are possible the bins are the aligned ranges of 2^k vertices, k the largest of
  // 13 .. 16 that still gives >= 64 bins (enough to spread the scatter's LDS counters; LJ stand-in: k = 16, 74 bins), and the
  // scatter needs no granule table: bin = id >> k, offset = id & (2^k - 1), the sorted entry is the id itself
  // (bin_scatter2_block<.., UNI>).  Fewer, wider bins also mean longer runs per bin and batch in the scatter's copy-out: the
  // balanced cut below (224 bins of ~21 k vertices on the LJ stand-in) costs the scatter 97 / 111 us on the two fat levels,
  // 74 uniform bins 82 / 97 (profiles/r5_c7_kernel_times_by_bin_cut.txt).  A hub-heavy range is balanced by the sweep, which
  // cuts a bin with more than its share of a level's candidates into parts.  GRX_BIN_UNIFORM=0: off; GRX_BIN_USHIFT=k: that width.
  int ushift = 0;
  // Which cut: measured both ways on two stand-ins (profiles/r5_c11_kernel_times_delta_table_vs_uniform.txt).  On the LJ stand-in
  // (14 edges per vertex) the uniform bins save the scatter 10 / 1 us per fat level and cost the sweep 20 / 14: its items are
  // the parts of 74 fat bins instead of ~214 whole bins of equal capacity -- a third larger at the top, and a part of a
  // 65536-vertex bin can win more vertices than its list holds.  On the kron stand-in (87 edges per vertex) the scatter is
  // nearly all of a fat level and the uniform cut wins 13 % of the search (697 -> 611 us).  So: uniform bins for graphs of
  // >= 32 edges per vertex, the balanced cut below otherwise; GRX_BIN_UNIFORM=0 | 1 forces one.
  const int uni_env = env_int("GRX_BIN_UNIFORM", -1);
  if (shift_max == 16 && (uni_env > 0 || (uni_env < 0 && (long long)g->E >= 32ll * g->V))) {
    int k = 16;
    while (k > 13 && k > gshift && (((long long)g->V + (1ll << k) - 1) >> k) < 64) --k;
    const int forced = env_int("GRX_BIN_USHIFT", 0);
    if (forced >= gshift && forced >= 10 && forced <= 16) k = forced;
    const long long nb_u = ((long long)g->V + (1ll << k) - 1) >> k;
    if (k >= gshift && nb_u >= 48 && nb_u <= BIN_MAX - 32) {
      ushift = k;
      for (int i = 0; i < n_gran; i += 1 << (k - gshift)) first.push_back(i);
    }
  }
  const bool uniform = ushift > 0;
  for (int attempt = 0; attempt < 64 && !uniform; ++attempt) {
    first.clear();
    long long acc = 0;
    int width = 0;
    for (int i = 0; i < n_gran; ++i) {
      if (width == 0) first.push_back(i);
      acc += cnt[(size_t)i];
      ++width;
      if (acc >= target || width == max_width) { acc = 0; width = 0; }
    }
    if ((int)first.size() <= BIN_MAX) break;
    target += target / 4 + 1;
  }
  const int nb = (int)first.size();
  if (nb < 1 || nb > BIN_MAX) return state.done(2);
  first.push_back(n_gran);
  std::vector<unsigned char> g2b((size_t)BIN_GRAN_MAX, 0), owner((size_t)BIN_MAX, 0);
  std::vector<int32_t> off((size_t)BIN_MAX + 1, 0), v0((size_t)BIN_MAX + 1, 0);
  std::vector<long long> cap((size_t)nb, 0);
  for (int b = 0; b < nb; ++b) {
    for (int i = first[(size_t)b]; i < first[(size_t)b + 1]; ++i) {
      g2b[(size_t)i] = (unsigned char)b;
      cap[(size_t)b] += cnt[(size_t)i];
    }
    v0[(size_t)b] = (int32_t)((long long)first[(size_t)b] << gshift);
  }
  {
    long long acc = 0;
    for (int b = 0; b <= BIN_MAX; ++b) {
      off[(size_t)b] = (int32_t)acc;
      if (b < nb) acc += cap[(size_t)b];
      if (b >= nb) v0[(size_t)b] = (int32_t)((long long)n_gran << gshift);  // bitmap words exist up to the padded end
    }
  }
  // owner: heaviest bin first, each to the least loaded XCD
  std::vector<int> order((size_t)nb);
  for (int b = 0; b < nb; ++b) order[(size_t)b] = b;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cap[(size_t)x] > cap[(size_t)y]; });
  std::vector<long long> load((size_t)ctx->n_xcd, 0);
  for (int b : order) {
    int best = 0;
    for (int x = 1; x < ctx->n_xcd; ++x)
      if (load[(size_t)x] < load[(size_t)best]) best = x;
    owner[(size_t)b] = (unsigned char)best;
    load[(size_t)best] += cap[(size_t)b] + 1;
  }
  // second scatter: granule -> bin | (index of the granule inside its bin) << 8, one 16-bit LDS read per edge
  std::vector<unsigned short> g2b16((size_t)BIN_GRAN_MAX, 0);
  for (int b = 0; b < nb; ++b)
    for (int i = first[(size_t)b]; i < first[(size_t)b + 1]; ++i)
      g2b16[(size_t)i] = (unsigned short)(b | ((i - first[(size_t)b]) << 8));
  static_assert((1 << (BIN_SHIFT_MAX - BIN_GSHIFT_MIN)) <= 256, "granule index inside a bin fits 8 bits");
  const size_t tab_bytes = (size_t)BIN_GRAN_MAX + (size_t)BIN_MAX + (size_t)BIN_GRAN_MAX * sizeof(unsigned short);  // g2b, owner, g2b16
  dev_scratch tab8, offs;  // adopted by the handle once everything has arrived
  GRX_HIP(tab8.alloc(tab_bytes));
  GRX_HIP(offs.alloc(2 * ((size_t)BIN_MAX + 1) * sizeof(int32_t)));  // off, v0
  unsigned char* d_tab8 = tab8.as<unsigned char>();
  int32_t* d_off = offs.as<int32_t>();
  GRX_HIP(hipMemcpyAsync(d_tab8, g2b.data(), (size_t)BIN_GRAN_MAX, hipMemcpyHostToDevice, s));
  GRX_HIP(hipMemcpyAsync(d_tab8 + BIN_GRAN_MAX, owner.data(), (size_t)BIN_MAX, hipMemcpyHostToDevice, s));
  static_assert((BIN_GRAN_MAX + BIN_MAX) % 4 == 0, "the 16-bit table is read as 32-bit words");
  GRX_HIP(hipMemcpyAsync(d_tab8 + BIN_GRAN_MAX + BIN_MAX, g2b16.data(), (size_t)BIN_GRAN_MAX * sizeof(unsigned short),
                         hipMemcpyHostToDevice, s));
  GRX_HIP(hipMemcpyAsync(d_off, off.data(), ((size_t)BIN_MAX + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  GRX_HIP(hipMemcpyAsync(d_off + BIN_MAX + 1, v0.data(), ((size_t)BIN_MAX + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  GRX_HIP(hipStreamSynchronize(s));
  g->bin_tab8 = reinterpret_cast<unsigned char*>(tab8.release());
  g->bin_off = reinterpret_cast<int32_t*>(offs.release());
  g->bin_shift = gshift;
  g->bin_ngran = n_gran;
  g->bin_nb = nb;
  g->bin_entry16 = shift_max == 16 ? 1 : 0;
  g->bin_uniform = ushift;
  return state.done(1);
}

// Workgroups of the per-level kernel: exactly what is RESIDENT (persistent workgroups
// stride over the work with gridDim, so a partial second round would double the time),
// one per CU on road-like graphs (advance_grid_for).

// The per-level kernel comes in two builds: bottom-up chunks in flight per wave
// (GRX_BU_BATCH = 2 | 4; measured equal on the LJ / kron stand-ins, 8 slower).  Forcing more
// waves per SIMD through __launch_bounds__ only produced spills (measured slower).
using level_kernel_fn = void (*)(pipe_args, dobfs_args, bfs_policy);
struct level_build {
  level_kernel_fn fn;
  int per_cu;  // resident workgroups per CU (0: not queried yet)
  int limit;   // ... and how many of them are used at most
};
static level_build* level_kernel_build() {
  static level_build builds[2] = {{bfs_level_kernel<2>, 0, 8}, {bfs_level_kernel<4>, 0, 8}};
  return &builds[env_int("GRX_BU_BATCH", 2) == 4 ? 1 : 0];
}
// second bottom-up body (grx_bfs_kernels.hpp): one round trip per round, unsettled lanes deferred.  GRX_BU2=0: first version
static level_build* level_kernel_build2(bool debug) {
  // 7 workgroups fit a CU; 4 measured fastest on all three scale-free stand-ins (profiles/history/r3_ab_bottomup_second_body.txt)
  static level_build builds[2] = {{bfs_level_kernel<2, true>, 0, 4}, {bfs_level_kernel<2, true, true>, 0, 4}};
  return &builds[debug ? 1 : 0];
}
using bin_kernel_fn = void (*)(pipe_args, bin_args, bfs_policy);
static int resident_per_cu(bin_kernel_fn fn) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, ADV_BLOCK, 0) != hipSuccess || n < 1) n = 2;
  return n > 8 ? 8 : n;
}

// Workgroups of the per-level kernel: exactly what is RESIDENT (persistent workgroups
// stride over the work with gridDim, so a partial second round would double the time),
// one per CU on road-like graphs (advance_grid_for).
static int level_grid(grx_context_t ctx, grx_graph_t g, level_build* lb) {
  if (lb->per_cu == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, lb->fn, ADV_BLOCK, 0) != hipSuccess || n < 1) n = 4;
    lb->per_cu = n > 8 ? 8 : n;
  }
  const int cap = env_int("GRX_LEVEL_WG_PER_CU", 0);  // tuning knob: another number of resident workgroups
  const int use = (cap > 0 && cap <= lb->per_cu) ? cap : (lb->per_cu < lb->limit ? lb->per_cu : lb->limit);
  const int full = advance_grid_for(ctx, g);
  const int resident = ctx->num_cus * use;
  return full < resident ? full : resident;
}

// tuning aid: the control block's spare counters as they are on the device now (GRX_MID_DEBUG=1: per-phase clock sums
// of the multi-level body, grx_mid.hpp)
extern "C" grx_status_t grx_debug_ctrl(grx_context_t ctx, int32_t* out, int32_t n) {
  if (!ctx || !out || n < 1 || n > 13) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_debug_ctrl: 1..13 counters");
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  ctrl_t h;
  GRX_HIP(hipMemcpy(&h, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) out[i] = i < 5 ? h.spare[i] : h.dbg_fine[i - 5];
  return GRX_SUCCESS;
}

// tuning aid: copy `n` 64-bit words of the context's debug scratch (GRX_BIN_DEBUG) to the host
extern "C" grx_status_t grx_debug_read(grx_context_t ctx, long long* out, int64_t n) {
  if (!ctx || !out || !ctx->far[1].ptr || (size_t)n * sizeof(long long) > ctx->far[1].bytes)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_debug_read: nothing recorded");
  GRX_HIP(hipMemcpy(out, ctx->far[1].ptr, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
  return GRX_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------
// One BFS, as an object: everything grx_bfs() sets up once per search and the launch group it enqueues per level.  The
// single-GPU entry point drives it with run_levels(); the partitioned entry points (grx_bfs_dist_*, below) drive the SAME
// object with the exchange between launch_group() and launch_post() -- at one rank the two are the same code path.
// ---------------------------------------------------------------------------------------------------------------
// what grx_bfs_dist_create fixed for a partitioned handle
struct bfs_part_cfg {
  part_args x{};               // (sent / do_run are filled per search)
  grx_graph_t g_in = nullptr;  // in-rows of the owned slice (null: the out-rows of a symmetric graph, or no direction switch)
};

static const grx_status_t BFS_RETRY_STATIC = static_cast<grx_status_t>(1000);  // internal: repeat the search with statically strided scatter units

struct bfs_search {
  grx_context_t ctx = nullptr;
  grx_graph_t g = nullptr;
  grx_options_t opt{};
  int32_t src = 0;
  int32_t* d_dist = nullptr;
  bool part = false;        // more than one rank: partition kernels, exchange between launch_group and launch_post
  bool api_driven = false;  // the level groups are enqueued by the caller (grx_bfs_dist_pre / _post): generic groups, no hints
  bool answered = false;    // setup() ran the whole search on another path (block-asynchronous relaxation)

  pipe_args a{};
  dobfs_args d{};
  bfs_policy lp{};
  bfs_policy_part lpp{};
  bin_args bn{};
  part_args x{};
  int variant = 0;
  bool dopt = false, strict_mp = false, fwd_bm = false, use_bins = false, profile = false, dense = false;
  level_build* lbuild = nullptr;
  bool part_bu2 = false;
  unsigned* visited = nullptr;
  int grid = 0, grid_scatter = 0, grid_scatter2 = 0, grid_sweep3 = 0, grid_post = 0;
  int pace = 0, hold_after = 0;
  uint32_t bin_groups = ~0u, exact_groups = 0u;
  bool exact = false, ended_in_head = false, do_repeat = false;
  hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
  hipError_t launch_err = hipSuccess;
  int launches = 0;
  int64_t prof_v = 0, prof_e = 0, prof_open = 0, prof_probe = 0;

  grx_status_t setup(grx_context_t ctx_, grx_graph_t g_, int32_t src_, const grx_options_t* options, int32_t* d_dist_,
                     const bfs_part_cfg* pc, bool api_driven_, float* elapsed_ms);
  void launch_group(hipStream_t stream, int seq);
  void launch_post(hipStream_t stream);
  void after_sync(const ctrl_t& h);
  grx_status_t finish(bool returned_fast, int groups_used, float* elapsed_ms);
  void drop_events() {
    for (auto& e : pe)
      if (e) { (void)hipEventDestroy(e); e = nullptr; }
  }
};

using part_level_fn = void (*)(pipe_args, dobfs_args, bfs_policy_part, int);
// (`full`: the widest grid the graph is given -- decided by the density of the WHOLE graph, not of the rows a rank holds)
static int part_level_grid(grx_context_t ctx, int full, part_level_fn fn, int limit) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, ADV_BLOCK, 0) != hipSuccess || n < 1) n = 4;
  if (n > limit) n = limit;
  const int resident = ctx->num_cus * n;
  return full < resident ? full : resident;
}

grx_status_t bfs_search::setup(grx_context_t ctx_, grx_graph_t g_, int32_t src_, const grx_options_t* options, int32_t* d_dist_,
                               const bfs_part_cfg* pc, bool api_driven_, float* elapsed_ms) {
  ctx = ctx_;
  g = g_;
  src = src_;
  d_dist = d_dist_;
  api_driven = api_driven_;
  part = pc != nullptr && pc->x.P > 1;
  if (pc) x = pc->x;
  if (options) opt = *options; else grx_options_default(&opt);
  if (opt.advance_load_balance == GRX_LB_WORK_STEALING)
    return fail(GRX_ERROR_UNSUPPORTED, "Load balance type not supported.");

  GRX_HIP(hipSetDevice(ctx->device));
  ctx->block_stats = grx_block_stats_t{};
  variant = part ? 0 : (opt.engine_flags >> 8) & 7;
  grx_status_t st;
  // road-like graphs: block-asynchronous relaxation (grx_block.hip) instead of thousands of nearly empty levels
  if (!part && !api_driven && variant == 0 && opt.max_iterations == 0 &&
      !(opt.engine_flags & (GRX_FLAG_NO_BLOCK_ASYNC | GRX_FLAG_UNFUSED | GRX_FLAG_SYNC_EACH_LEVEL | GRX_FLAG_LB_STRICT)) &&
      env_int("GRX_LB_STRICT", 0) == 0) {
    bool use = false;
    st = blk_prepare(ctx, g, false, &use);
    if (st != GRX_SUCCESS && !build_failed_softly(st)) return st;
    if (st == GRX_SUCCESS && use) {
      answered = true;
      return blk_search(ctx, g, src, opt, false, d_dist, elapsed_ms);
    }
  }
  st = pipeline_prepare(ctx, g, &a);
  if (st != GRX_SUCCESS) return st;
  // edges of the WHOLE graph (a rank of a partition holds the rows of its slice only)
  const long long e_all = part ? x.e_global : (long long)g->E;
  // bottom-up pays off on low-diameter graphs; with fewer than 4 edges per vertex the
  // frontier never gets heavy enough to switch and the extra per-level kernels only cost
  dopt = opt.advance_direction == GRX_DIR_OPTIMIZED && variant == 0 && e_all >= 4ll * g->V;
  grx_graph_t gt = g;  // the handle whose arrays are the in-edges (and that caches what is derived from them)
  if (dopt && part) {
    // a partition brings the in-rows of its slice (or is symmetric: its out-rows are its in-rows)
    if (pc->g_in) gt = pc->g_in;
    else if (!g->symmetric) dopt = false;
  } else if (dopt) {
    // in-edges of the bottom-up step: the cached transpose (built on first use).  When it cannot be had -- the stable sort
    // needs 16-24 transient bytes per edge beside the 4-8 it keeps -- the search runs FORWARD-ONLY instead of failing: same
    // depths, the reference's own advance direction (the binned levels fall back the same way).
    st = graph_build_transpose(ctx, g);
    if (st == GRX_ERROR_OUT_OF_MEMORY) {
      (void)hipGetLastError();
      dopt = false;
    } else if (st != GRX_SUCCESS) {
      return st;
    }
  }
  const size_t bm_words = 4 * (((size_t)g->V + 127) / 128);  // whole 16-byte groups: 64-vertex chunks, uint4 clears
  hipStream_t s = ctx->stream;
  // widest per-level grid (advance_grid_for: one workgroup per CU on road-like graphs) -- by the density of the whole graph
  const int grid_full = part ? ((g->V > 0 && e_all < 4ll * g->V) ? ctx->num_cus : ctx->num_cus * 8) : advance_grid_for(ctx, g);
  // Forward-only runs on dense graphs keep a visited bitmap too: it pre-filters the label probes of
  // the claim-per-edge advance and carries the binned fat levels (grx_bin.hpp).
  // Tuning knobs: GRX_TD_BITMAP=0 (no bitmap at all), GRX_TD_PRE=0 (no pre-filter), GRX_TD_BIN=0 (no
  // binned levels), GRX_BIN_MIN_EDGES (out-edges of a frontier from which a level is binned).
  // GRX_FLAG_LB_STRICT + merge_path: every level on the chunked merge-path advance, nothing else
  strict_mp = !part && ((opt.engine_flags & GRX_FLAG_LB_STRICT) != 0 || env_int("GRX_LB_STRICT", 0) != 0) &&
              (opt.advance_load_balance == GRX_LB_MERGE_PATH || opt.advance_load_balance == GRX_LB_MERGE_PATH_V2);
  // (a partitioned forward run always keeps the bitmap: its bits for other ranks' vertices are the record of what was reported)
  fwd_bm = !dopt && !strict_mp && variant == 0 && (part || (e_all >= 4ll * g->V && env_int("GRX_TD_BITMAP", 1) != 0));
  use_bins = fwd_bm && e_all >= 4ll * g->V && env_int("GRX_TD_BIN", 1) != 0;
  if (use_bins) {
    st = graph_build_bins(ctx, g);
    if (st != GRX_SUCCESS && !build_failed_softly(st)) return st;
    use_bins = st == GRX_SUCCESS && g->bin_state == 1;  // (no room for the tables: fat levels on the claim-per-edge advance)
  }
  if (use_bins) {
    // per-SEARCH scratch of the binned levels, owned by the context: the E-entry candidate array (+ the tail of a 16-byte
    // load) and the fill / ticket words.  When the big array cannot be had (4 E bytes: 2 GB on the 530 M-edge graph) the
    // search runs without binned levels -- slower fat levels, same result -- instead of failing.
    const size_t fill_bytes = ((size_t)BIN_MAX + 16) * BIN_PAD * sizeof(int32_t);
    const bool fresh = ctx->bin_fill.bytes < fill_bytes;
    if (ctx->bins.reserve(((size_t)g->E + 16) * sizeof(int32_t)) != hipSuccess || ctx->bin_fill.reserve(fill_bytes) != hipSuccess) {
      (void)hipGetLastError();
      use_bins = false;
    } else if (fresh) {
      GRX_HIP(hipMemsetAsync(ctx->bin_fill.ptr, 0, ctx->bin_fill.bytes, ctx->stream));
    }
  }

  d = dobfs_args{};
  d.dist = d_dist;
  d.n_words = (int32_t)bm_words;
  d.n_edges = g->E;
  d.enabled = dopt ? 1 : 0;
  d.alpha = env_int("GRX_DO_ALPHA", DO_ALPHA);
  d.beta = env_int("GRX_DO_BETA", DO_BETA);
  d.back_div = env_int("GRX_DO_BACK_DIV", 0);
  if (d.alpha < 1) d.alpha = 1;
  if (d.beta < 1) d.beta = 1;
  lbuild = level_kernel_build();
  visited = nullptr;
  size_t visited_bytes = 0;
  if (dopt) {
    // in-edges: the CSR itself when the graph is symmetric (the property is caller-supplied and
    // defaults to true, so it is verified once per graph handle), else the cached transpose
    // Round 4: always the transpose -- built by the stable sort in a few ms, its in-lists hubs first (the likeliest
    // parents are probed first).  GRX_BU_SYMMETRIC_CSR=1: a graph declared symmetric uses its own CSR instead (rounds 1-3),
    // after the claim has been verified against the transpose.
    bool use_csr = false;
    if (!part && g->symmetric && env_int("GRX_BU_SYMMETRIC_CSR", 0) != 0) {
      st = graph_is_symmetric(ctx, g, &use_csr);
      if (st != GRX_SUCCESS) return st;
    }
    if (part) {
      // the in-rows the partition brought, their hub sources first (built once per handle; without it: as they came)
      st = graph_build_hub_first(ctx, gt);
      if (st != GRX_SUCCESS) return st;
      d.t_ro = gt->ro;
      d.t_ci = gt->hf_state == 1 ? gt->hf_ci : gt->ci;
    } else {
      d.t_ro = use_csr ? g->ro : g->t_ro;
      d.t_ci = use_csr ? g->ci : g->t_ci;
    }
    GRX_HIP(ctx->bitmap[0].reserve(bm_words * sizeof(unsigned)));
    GRX_HIP(ctx->bitmap[1].reserve(3 * bm_words * sizeof(unsigned)));
    d.visited = ctx->bitmap[0].as<unsigned>();
    d.fbits[0] = ctx->bitmap[1].as<unsigned>();
    d.fbits[1] = d.fbits[0] + bm_words;
    d.fbits[2] = d.fbits[1] + bm_words;
    d.rot3 = 1;
    {
      // The two per-graph arrays below are built on first use: into LOCALS, by kernels on this context's stream, and published
      // under the handle's build lock only after that stream has drained (a second context on another stream that
      // takes the lock next must find them BUILT, not merely allocated -- the transpose, bin and PageRank builders do the same).
      std::lock_guard<std::recursive_mutex> lk(gt->prep_mu);
      const bool want_heads = env_int("GRX_BU_HEADS", 1) != 0 && g->V < (1 << 29);
      const bool build_closed = !gt->closed0 || gt->closed0_words != (int32_t)bm_words || gt->closed0_of != (const void*)d.t_ro;
      const bool build_heads = want_heads && (!gt->bu_heads || gt->bu_heads_of != (const void*)d.t_ci);
      if (build_closed || build_heads) {
        dev_scratch closed_new, heads_new;
        if (build_closed) {
          // static "no in-edges" bitmap, built once per graph and kept in the graph handle
          GRX_HIP(closed_new.alloc(bm_words * sizeof(unsigned)));
          hipLaunchKernelGGL(bfs_closed0_kernel, dim3(ctx->num_cus * 4), dim3(256), 0, s, d.t_ro, g->V, closed_new.as<unsigned>(),
                             (int)bm_words);
        }
        if (build_heads) {
          // the first two in-neighbours of every vertex as one dense array, built once per graph (8 V bytes)
          GRX_HIP(heads_new.alloc((size_t)g->V * 2 * sizeof(int32_t)));
          hipLaunchKernelGGL(bfs_heads_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, s, d.t_ro, d.t_ci, g->V, heads_new.as<int2>());
        }
        GRX_HIP(hipGetLastError());
        GRX_HIP(hipStreamSynchronize(s));
        if (build_closed) {
          if (gt->closed0) GRX_HIP(hipFree(gt->closed0));
          gt->closed0 = reinterpret_cast<unsigned*>(closed_new.release());
          gt->closed0_words = (int32_t)bm_words;
          gt->closed0_of = (const void*)d.t_ro;
        }
        if (build_heads) {
          if (gt->bu_heads) GRX_HIP(hipFree(gt->bu_heads));
          gt->bu_heads = reinterpret_cast<int32_t*>(heads_new.release());
          gt->bu_heads_of = (const void*)d.t_ci;
        }
      }
      if (want_heads) d.heads = reinterpret_cast<const int2*>(gt->bu_heads);
    }
    if (part) {
      // the partition's builds of the level kernel; the second bottom-up body under the same condition as below
      part_bu2 = false;
      if (d.heads && env_int("GRX_BU2", 1) != 0) {
        const int grid2 = part_level_grid(ctx, grid_full, bfs_level_part_kernel<2, true>, 4);
        const long long chunks = (long long)bm_words / 2, per_round = (long long)grid2 * (ADV_BLOCK / 64) * 2;
        if ((chunks + per_round - 1) / per_round * 2 <= 128) part_bu2 = true;
      }
      d.bu_grid = part_bu2 ? part_level_grid(ctx, grid_full, bfs_level_part_kernel<2, true>, 4)
                           : part_level_grid(ctx, grid_full, bfs_level_part_kernel<2, false>, 8);
    } else {
      if (d.heads && env_int("GRX_BU2", 1) != 0) {
        // second bottom-up body: needs the dense array and all chunks of a wave in 128 slots
        d.debug_level = env_int("GRX_BU_DEBUG", 0);  // per-wave phase clocks of that level (tools/bu_debug.py)
        level_build* lb2 = level_kernel_build2(d.debug_level != 0);
        if (d.debug_level != 0) {
          GRX_HIP(ctx->far[1].reserve((size_t)8 * 16384 * sizeof(long long)));
          d.debug = ctx->far[1].as<long long>();
          GRX_HIP(hipMemsetAsync(d.debug, 0, (size_t)8 * 16384 * sizeof(long long), s));
        }
        const int grid2 = level_grid(ctx, g, lb2);
        const long long chunks = (long long)bm_words / 2, per_round = (long long)grid2 * (ADV_BLOCK / 64) * 2;
        if ((chunks + per_round - 1) / per_round * 2 <= 128) lbuild = lb2;
      }
      d.bu_grid = level_grid(ctx, g, lbuild);
    }
    GRX_HIP(ctx->bu_part.reserve((size_t)d.bu_grid * 4 * sizeof(long long)));
    d.bu_part = ctx->bu_part.as<long long>();
    a.bu_part = d.bu_part;
  } else if (fwd_bm || (variant != 0 && variant != 7)) {
    visited_bytes = bm_words * sizeof(unsigned);
    GRX_HIP(ctx->bitmap[0].reserve(visited_bytes));
    visited = ctx->bitmap[0].as<unsigned>();
  }
  if (part) {
    // what this rank has reported: the visited bitmap itself in a forward run, a bitmap of its own beside the three frontier
    // bitmaps of a direction-optimising one (whose `visited` counts every vertex without local in-edges as closed)
    x.do_run = dopt ? 1 : 0;
    if (dopt) {
      GRX_HIP(ctx->labels.reserve((size_t)x.P * x.slice_words * sizeof(unsigned)));
      x.sent = ctx->labels.as<unsigned>();
      d.fin_global = x.recv;
    } else {
      x.sent = visited;
    }
  }

  // problem.reset() -- outside the timed region, as in the reference; direction-optimising runs: one launch with the seed
  // (bfs_reset_seed_kernel, below: inside the timed region then).  GRX_SEED_IN_RESET=0: two launches
  const bool seed_in_reset = !part && dopt && variant == 0 && env_int("GRX_SEED_IN_RESET", 1) != 0;
  // forward-only runs with a visited bitmap: reset + seed in one launch too (bfs_fwd_reset_seed_kernel); GRX_FWD_SEED_IN_RESET=0:
  // fill + memset + seed kernel
  const bool fwd_seed_in_reset = !part && !dopt && variant == 0 && fwd_bm && visited != nullptr && (bm_words & 3) == 0 &&
                                 (reinterpret_cast<uintptr_t>(d_dist) & 15) == 0 && env_int("GRX_FWD_SEED_IN_RESET", 1) != 0;
  if (part || seed_in_reset || fwd_seed_in_reset) {
  } else if (dopt) {
    hipLaunchKernelGGL(bfs_reset_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, s, d_dist, (int64_t)g->V, d, g->closed0);
  } else {
    GRX_HIP(fill_i32(s, d_dist, INT_MAX, g->V));
    if (visited) GRX_HIP(hipMemsetAsync(visited, 0, visited_bytes, s));
  }

  dense = g->V > 0 && e_all >= 8ll * g->V;  // few fat levels: paced enqueueing
  ctx->h_mailbox[0] = 0;
  ctx->h_mailbox[3] = -1;
  GRX_HIP(hipEventRecord(ctx->ev_begin, s));

  if (part) grid = dopt ? d.bu_grid : 0;
  else grid = (variant == 0) ? level_grid(ctx, g, lbuild) : advance_grid_for(ctx, g);
  profile = !part && !api_driven && (opt.engine_flags & GRX_FLAG_PROFILE) != 0;
  ctx->levels.clear();
  if (profile) for (auto& e : pe) GRX_HIP(hipEventCreate(&e));

  lp = bfs_policy{};
  lp.dist = d_dist;
  if (dopt) {
    lp.bm_visited = d.visited;
    for (int i = 0; i < 3; ++i) lp.bm_f[i] = d.fbits[i];
    lp.bm_words = d.n_words;
  } else if (fwd_bm) {
    lp.bm_visited = visited;
    for (int i = 0; i < 3; ++i) lp.bm_f[i] = visited;  // every discovery sets its bit in `visited` itself
    lp.bm_words = (int)bm_words;
    lp.fwd_bitmap = 1;
    // the read-only pre-filter of the label probe measured SLOWER (LJ stand-in, 31 M-edge level 525 -> 802 us:
    // a second dependent round trip per edge costs more than the label sectors it saves): off by default
    lp.pre_bm = (!part && env_int("GRX_TD_PRE", 0) != 0) ? visited : nullptr;
  }
  if (part) {
    lpp = bfs_policy_part{};
    static_cast<bfs_policy&>(lpp) = lp;
    lpp.lo = x.lo;
    lpp.hi = x.hi;
    lpp.sent = x.sent;
    lpp.send = x.send;
  }
  // seed; then level 0 itself when the source is a hub (bfs_source_kernel: a no-op otherwise).  Profiled and
  // strict-merge-path runs keep one launch pair per level, level 0 included.
  // (2: the source kernel also writes level 1's chunk map and counters -- forward runs with binned levels, whose head takes
  // them from the producers; GRX_SOURCE_MAP=0: the head of level 1 walks the tiles)
  const int source_level = (!part && variant == 0 && !profile && !strict_mp && env_int("GRX_SOURCE_LEVEL", 1) != 0)
                               ? ((use_bins && !dopt && env_int("GRX_SOURCE_MAP", 1) != 0) ? 2 : 1) : 0;
  static_assert(TILE == 256, "the seed runs in a workgroup of the reset kernel");
  if (part)
    hipLaunchKernelGGL(bfs_part_reset_seed_kernel, dim3(ctx->num_cus * 8), dim3(TILE), 0, s, d_dist, a, d, x,
                       dopt ? d.visited : visited, (int)bm_words, dopt ? gt->closed0 : nullptr, src);
  else if (seed_in_reset)
    hipLaunchKernelGGL(bfs_reset_seed_kernel, dim3(ctx->num_cus * 8), dim3(TILE), 0, s, d_dist, (int64_t)g->V, d, g->closed0, a, src,
                       source_level);
  else if (fwd_seed_in_reset)
    hipLaunchKernelGGL(bfs_fwd_reset_seed_kernel, dim3(ctx->num_cus * env_int("GRX_RESET_WG_PER_CU", 8)), dim3(TILE), 0, s, d_dist, (int64_t)g->V, visited,
                       (int)bm_words, a, src, d, source_level);
  else
    hipLaunchKernelGGL(bfs_init_kernel, dim3(1), dim3(TILE), 0, s, a, d_dist, visited, src, d, source_level);
  if (source_level)
    hipLaunchKernelGGL(bfs_source_kernel, dim3(ctx->num_cus * env_int("GRX_SOURCE_WG_PER_CU", 4)), dim3(ADV_BLOCK), 0, s, a, d, lp, src,
                       source_level == 2 ? 1 : 0);
  bn = bin_args{};
  bn.xcc_mask = ctx->xcc_mask;
  bn.n_xcd = ctx->n_xcd;
  d.xcc_mask = ctx->xcc_mask;
  // forward-only runs: frontiers of a few thousand vertices run many levels per launch (grx_mid.hpp); GRX_MID=0: off
  // (not in a partition: a level ends with an exchange)
  if (!part && !dopt && !strict_mp && variant == 0 && env_int("GRX_MID", 1) != 0) {
    bn.mid_v = env_int("GRX_MID_V", MID_ENTER_V);
    bn.mid_e = env_int("GRX_MID_E", MID_ENTER_E);
    bn.mid_tile_e = env_int("GRX_MID_TILE_E", MID_TILE_E);  // 0: frontiers with heavy tiles enter that body too (before round 5's last session)
  }
  grid_scatter = grid_scatter2 = grid_sweep3 = 0;
  if (use_bins) {
    bn.bins = ctx->bins.as<int32_t>();
    bn.off = g->bin_off;
    bn.v0 = g->bin_off + BIN_MAX + 1;
    bn.fill = ctx->bin_fill.as<int32_t>();
    bn.queue = ctx->bin_fill.as<int32_t>() + (size_t)BIN_MAX * BIN_PAD;
    bn.g2b16 = reinterpret_cast<const unsigned short*>(g->bin_tab8 + BIN_GRAN_MAX + BIN_MAX);
    bn.gshift = g->bin_shift;
    bn.n_gran = g->bin_ngran;
    bn.nb = g->bin_nb;
    bn.uniform = g->bin_uniform;
    bn.sweep_balance = env_int("GRX_SW2_BALANCE", 0);  // measured slower (profiles/r5_c7_kernel_times_by_bin_cut.txt): off
    bn.xcc_mask = ctx->xcc_mask;
    bn.n_xcd = ctx->n_xcd;
    bn.max_degree = env_int("GRX_BIN_MAX_DEGREE", 0);  // (a limit on the frontier's mean out-degree for binning: none)
    bn.debug_level = part ? 0 : env_int("GRX_BIN_DEBUG", 0);
    if (bn.debug_level != 0) {
      // per-workgroup records of the LAST binned level: scatter at [0, 4096), claim at [4096 + grid, ...); read
      // back with grx_debug_read
      GRX_HIP(ctx->far[1].reserve((size_t)8 * 16384 * sizeof(long long)));
      bn.debug = ctx->far[1].as<long long>();
      GRX_HIP(hipMemsetAsync(bn.debug, 0, (size_t)8 * 16384 * sizeof(long long), s));
    }
    // (round 5: 2^21, was 2^20 -- the 1.29 M-edge fourth level of the deep stand-in costs scatter 24 + sweep 55 us binned and
    // 44 us on the claim-per-edge advance: profiles/r5_c15_kernel_sequences.txt)
    bn.min_edges = (long long)env_int("GRX_BIN_MIN_EDGES", 1 << 21);
    if (bn.min_edges < 1) bn.min_edges = 1;
    // ... and EARLY levels (less than a quarter of the graph visited) from half of that: they discover a third of their targets,
    // which is what the claim-per-edge body pays for (plan_in::bin_early_div).  GRX_BIN_EARLY_DIV=1: one threshold
    bn.bin_early_div = env_int("GRX_BIN_EARLY_DIV", 2);
    if (bn.bin_early_div < 1) bn.bin_early_div = 1;
    bn.visited = visited;
    bn.visited_words = (int32_t)bm_words;
    bn.dist = d_dist;
    if (part) {
      bn.part_send = x.send;
      bn.part_wlo = x.lo >> 5;
      bn.part_whi = (x.hi + 31) >> 5;
    }
    {
      // scatter: 1024-thread workgroups, two per CU (the bins hold offsets inside the bin)
      static const int per_cu_sc2 = [] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bfs_scatter2_kernel<false, false>, SC2_BLOCK, 0) != hipSuccess || n < 1) n = 1;
        return n > 2 ? 2 : n;
      }();
      bn.local_ids = 1;
      grid_scatter2 = ctx->num_cus * env_int("GRX_SC2_WG_PER_CU", per_cu_sc2);
      if (grid_scatter2 < 1) grid_scatter2 = ctx->num_cus;
      // per-XCD ticket queues need a workgroup on every XCD: a grid that cannot be trusted to give that (fewer than four
      // workgroups per XCD), a context whose coverage check failed once, or GRX_SC2_STATIC=1 -> statically strided units
      bn.static_units = (ctx->sc2_static || grid_scatter2 < 4 * ctx->n_xcd || env_int("GRX_SC2_STATIC", 0) != 0) ? 1 : 0;
      bn.fault_xcd = ctx->sc2_static ? 0 : env_int("GRX_SC2_FAULT_XCD", 0);
      bn.sub_shift = env_int("GRX_BIN_SUB", 4) == 1 ? 0 : 2;
      // sweep: a bin with more than 1/160 of the level's candidates is cut into parts; the grid is twice the resident
      // workgroups (one per CU), so that a level cut into more items than CUs still runs
      grid_sweep3 = ctx->num_cus * 2;
      bn.sweep_items = env_int("GRX_SW2_ITEMS", bn.sweep_balance ? std::max(ctx->num_cus, bn.nb + 32) : bn.nb + 160);
      if (bn.sweep_items > grid_sweep3) bn.sweep_items = grid_sweep3;
    }
  }
  // 16-bit bin entries need the second scatter (writes them) and the second sweep (reads them)
  bn.entry16 = (use_bins && g->bin_entry16 && env_int("GRX_BIN_E16", 1) != 0) ? 1 : 0;  // (GRX_BIN_E16=0: 32-bit entries in the same bins, for A/B)
  if (!dopt && variant == 0) {
    static const int per_cu_scatter = resident_per_cu(bfs_level_bin_kernel);
    const int resident = ctx->num_cus * per_cu_scatter;
    const int full = grid_full;  // one workgroup per CU on road-like graphs
    grid_scatter = full < resident ? full : resident;
    if (const int k = env_int("GRX_LEVEL_WG_PER_CU", 0); k > 0) grid_scatter = std::min(grid_scatter, ctx->num_cus * k);  // (tuning aid)
  }
  if (part) {
    const int n_seg = ((x.hi - x.lo + 31) / 32 + PP_SEG_WORDS - 1) / PP_SEG_WORDS;
    grid_post = std::max(1, std::min(n_seg, ctx->num_cus * 2));
  }
  // thin top-down levels run on as many workgroups as their chunk count asks for (thin_workgroups, grx_bfs_kernels.hpp);
  // GRX_THIN_CHUNKS_PER_WG=0: every workgroup of the launch takes chunks
  bn.thin_div = d.thin_div = (variant == 0 && !strict_mp) ? env_int("GRX_THIN_CHUNKS_PER_WG", 1) : 0;
  bn.thin_min = d.thin_min = ctx->num_cus;
  launch_err = hipSuccess;
  launches = 0;
  pace = (dense && variant == 0 && !part && !api_driven) ? env_int("GRX_PACE_DEPTH", 2) : 0;
  // The scatter and sweep kernels of a binned level are launched blindly with every group and cost a launch each where the
  // level is thin (two no-op kernels, ~4 us each, on four of the six levels of the LJ stand-in).  The head kernels
  // record in which groups they met a fat level (ctrl_t::bin_want, published with `done`); the next forward search on the
  // graph leaves the two kernels out of the other groups, one group of slack either side.  A fat level in a group
  // without them runs on the claim-per-edge advance (the head is told: bin_args::allowed) and is recorded for the next
  // search.  GRX_BIN_HINT=0: every group carries them.
  // (Level groups a caller enqueues itself, and the groups of a partition -- every rank must enqueue the same kernels around
  // the same collectives -- are always the full generic group.)
  const bool hints = !part && !api_driven;
  const uint32_t hint0 = (hints && use_bins && env_int("GRX_BIN_HINT", 1) != 0) ? g->bin_hint.load(std::memory_order_relaxed) : 0u;
  bin_groups = hint0 ? (hint0 | (hint0 << 1) | (hint0 >> 1)) : ~0u;
  // The same source as the last forward search on this handle: the groups with a fat level are known EXACTLY (grx_graph::
  // bin_exact).  Those groups carry head + scatter + sweep, the others head + level kernel: a forward search on the LJ
  // stand-in spent ~25 us on six no-op launches (the level kernel in front of either fat level, scatter + sweep of the slack
  // group behind them).  A wrong prediction is only slow: a fat level in a group without the two kernels runs on the
  // claim-per-edge advance (bin_args::allowed), a thin one in a group without a level kernel is binned (bin_args::no_level).
  // GRX_BIN_EXACT=0: off.  Profiled runs (GRX_FLAG_PROFILE: one group per level, level 0 included) keep their own record, so
  // that the per-level times they report are those of the kernels a repeated search launches.
  exact_groups = 0u;
  exact = false;
  if (hints) {
    const uint64_t ex = g->bin_exact[profile ? 1 : 0].load(std::memory_order_relaxed);
    if (use_bins && hint0 && (uint32_t)(ex >> 32) == (uint32_t)src + 1u && env_int("GRX_BIN_EXACT", 1) != 0) {
      exact = true;
      exact_groups = (uint32_t)ex;
      bin_groups = exact_groups;
    }
  }
  // Paced searches: enqueue as many groups as the previous search on this graph (same direction rule) needed, then wait
  // for the end instead of queueing two more behind it (run_levels: hold_after).  GRX_GROUP_HINT=0: off
  // (the previous search on the handle, same direction rule, ended in a head kernel -- not in the many-levels body of a level kernel)
  ended_in_head = hints && g->end_in_head[dopt ? 1 : 0].load(std::memory_order_relaxed) != 0;
  do_repeat = hints && dopt && variant == 0 && g->do_last_src.load(std::memory_order_relaxed) == (uint32_t)src + 1u;
  hold_after = (pace > 0 && env_int("GRX_GROUP_HINT", 1) != 0) ? g->group_hint[dopt ? 1 : 0].load(std::memory_order_relaxed) : 0;
  if (hold_after > 0 && env_int("GRX_GROUP_HINT_FORCE", 0) > 0) hold_after = env_int("GRX_GROUP_HINT_FORCE", 0);  // (test aid: a wrong prediction)
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

// One level group.  Single GPU: everything of the level.  Partition: what precedes the exchange of the outgoing bitmap.
void bfs_search::launch_group(hipStream_t stream, int seq) {
  if (part) {
    bn.allowed = use_bins ? 1 : 0;
    bn.no_level = 0;
    bn.only_finish = 0;
    hipLaunchKernelGGL(bfs_head_part_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, a, d, seq, bn, x);
    if (!dopt) {
      hipLaunchKernelGGL(bfs_level_bin_part_kernel, dim3(grid_scatter), dim3(ADV_BLOCK), 0, stream, a, bn, lpp);
      if (use_bins) {
        if (bn.entry16 && bn.uniform)
          hipLaunchKernelGGL((bfs_scatter2_kernel<false, true, true>), dim3(grid_scatter2), dim3(SC2_BLOCK), 0, stream, a, bn);
        else if (bn.entry16)
          hipLaunchKernelGGL((bfs_scatter2_kernel<false, true>), dim3(grid_scatter2), dim3(SC2_BLOCK), 0, stream, a, bn);
        else
          hipLaunchKernelGGL((bfs_scatter2_kernel<false, false>), dim3(grid_scatter2), dim3(SC2_BLOCK), 0, stream, a, bn);
        if (bn.entry16)
          hipLaunchKernelGGL((bfs_sweep2_part_kernel<SW3_BLOCK, SW3_LIST, 4, true>), dim3(grid_sweep3), dim3(SW3_BLOCK), 0, stream, a, bn);
        else
          hipLaunchKernelGGL((bfs_sweep2_part_kernel<SW3_BLOCK, SW3_LIST, 4, false>), dim3(grid_sweep3), dim3(SW3_BLOCK), 0, stream, a, bn);
      }
    } else {
      hipLaunchKernelGGL(bfs_part_prep_kernel, dim3(ctx->num_cus * 2), dim3(256), 0, stream, a, d, x);
      if (part_bu2) hipLaunchKernelGGL((bfs_level_part_kernel<2, true>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a, d, lpp, 0);
      else hipLaunchKernelGGL((bfs_level_part_kernel<2, false>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a, d, lpp, 0);
    }
    ++launches;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) launch_err = e;
    return;
  }
  if (profile) (void)hipEventRecord(pe[0], stream);
  const bool bins_here = use_bins && (seq >= 32 || ((bin_groups >> seq) & 1u) != 0u);
  // (the second scatter + a sweep: the first versions of the two kernels share the level kernel's launch)
  // ... and the group in which the previous search from this source ENDED is its head alone when `done` was set by that head
  // (the tiny levels inside it, or its plan step finding the frontier empty; mailbox[12] says so -- the many-levels body of a
  // level kernel may end a search too): the level kernel behind that head was a 4 us no-op in front of the next search.  A search that does not end there after all is left untouched by that head
  // (plan_in::only_finish) and continues in the next group.  GRX_LAST_HEAD_ONLY=0: off
  // (direction-optimising searches: the same source as the last such search on the handle, grx_graph::do_last_src)
  // (not with a cap on the iterations: a head-only group would use up one of the caller's groups without advancing a level)
  const bool only_head = (exact || do_repeat) && ended_in_head && !profile && hold_after > 0 && seq == hold_after - 1 && !bins_here &&
                         opt.max_iterations == 0 && env_int("GRX_LAST_HEAD_ONLY", 1) != 0;
  const bool level_here = !(exact && bins_here && seq < 32) && !only_head;
  bn.allowed = bins_here ? 1 : 0;
  bn.no_level = (level_here || only_head) ? 0 : 1;
  bn.only_finish = only_head ? 1 : 0;
  if (variant == 0) {
    // head (tiny levels + decide + plan) -> level
    hipLaunchKernelGGL(bfs_head_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, a, d, lp, (profile || strict_mp) ? 0 : 1, seq, bn);
    if (profile) (void)hipEventRecord(pe[1], stream);
    if (!dopt) {
      // forward-only run.  level = claim-per-edge advance, many mid-size levels (grx_mid.hpp), or the SCATTER phase of
      // a binned level; the CLAIM phase is launched only when levels can be binned (a no-op unless the head did)
      if (level_here) hipLaunchKernelGGL(bfs_level_bin_kernel, dim3(grid_scatter), dim3(ADV_BLOCK), 0, stream, a, bn, lp);
      // (four builds of each: tuning clocks on / off x 16-bit / 32-bit bin entries)
      auto launch2 = [&](auto dbg_c, auto e16_c) {
        constexpr bool DBG = decltype(dbg_c)::value, E16 = decltype(e16_c)::value;
        if (grid_scatter2 > 0 && E16 && bn.uniform)
          hipLaunchKernelGGL((bfs_scatter2_kernel<DBG, E16, E16>), dim3(grid_scatter2), dim3(SC2_BLOCK), 0, stream, a, bn);
        else if (grid_scatter2 > 0)
          hipLaunchKernelGGL((bfs_scatter2_kernel<DBG, E16>), dim3(grid_scatter2), dim3(SC2_BLOCK), 0, stream, a, bn);
        if (use_bins)
          hipLaunchKernelGGL((bfs_sweep2_kernel<SW3_BLOCK, SW3_LIST, 4, DBG, E16>), dim3(grid_sweep3), dim3(SW3_BLOCK), 0, stream, a, bn);
      };
      using std::true_type;
      using std::false_type;
      if (!bins_here) { /* thin level expected */ }
      else if (bn.debug && bn.entry16) launch2(true_type{}, true_type{});
      else if (bn.debug) launch2(true_type{}, false_type{});
      else if (bn.entry16) launch2(false_type{}, true_type{});
      else launch2(false_type{}, false_type{});
    } else {
      if (!only_head) hipLaunchKernelGGL(lbuild->fn, dim3(grid), dim3(ADV_BLOCK), 0, stream, a, d, lp);
    }
  } else {
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, a, 0);
    if (profile) (void)hipEventRecord(pe[1], stream);
    auto adv = [&](auto pol) {
      hipLaunchKernelGGL((advance_kernel<decltype(pol)>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a, pol);
    };
    switch (variant) {
      case 1: { bfs_policy_t<1> q{}; q.dist = d_dist; q.visited = visited; adv(q); break; }
      case 2: { bfs_policy_t<2> q{}; q.dist = d_dist; q.visited = visited; adv(q); break; }
      case 3: { bfs_policy_t<3> q{}; q.dist = d_dist; q.visited = visited; adv(q); break; }
      default: { bfs_policy_t<7> q{}; q.dist = d_dist; adv(q); }
    }
  }
  ++launches;
  if (profile) {
    (void)hipEventRecord(pe[2], stream);
    (void)hipEventSynchronize(pe[2]);
    float t_plan = 0, t_adv = 0;
    (void)hipEventElapsedTime(&t_plan, pe[0], pe[1]);
    (void)hipEventElapsedTime(&t_adv, pe[1], pe[2]);
    level_rec r{};
    r.advance_ms = t_adv;
    r.other_ms = t_plan;
    ctx->levels.push_back(r);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) launch_err = e;
}

// Partition only: what follows the exchange -- a bottom-up level (it probes the whole-graph frontier the exchange assembled),
// the claim of what the peers reported, this rank's statistics of the next frontier (the caller all-reduces them).
void bfs_search::launch_post(hipStream_t stream) {
  if (!part) return;
  if (dopt) {
    if (part_bu2) hipLaunchKernelGGL((bfs_level_part_kernel<2, true>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a, d, lpp, 1);
    else hipLaunchKernelGGL((bfs_level_part_kernel<2, false>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a, d, lpp, 1);
  }
  hipLaunchKernelGGL(bfs_part_post_kernel, dim3(grid_post), dim3(PP_BLOCK), 0, stream, a, d, x, dopt ? d.visited : visited);
  hipLaunchKernelGGL(bfs_part_stats_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, a, d, x);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) launch_err = e;
}

void bfs_search::after_sync(const ctrl_t& h) {
  if (!profile || ctx->levels.empty()) return;
  // bottom-up open/probe counters are reduced by the NEXT level's decide kernel:
  // the delta seen now belongs to the previous level's record
  if (ctx->levels.size() >= 2) {
    level_rec& prev = ctx->levels[ctx->levels.size() - 2];
    prev.bu_open = h.bu_open - prof_open;
    prev.bu_probes = h.bu_probes - prof_probe;
  }
  prof_open = h.bu_open;
  prof_probe = h.bu_probes;
  level_rec& r = ctx->levels.back();
  r.frontier_size = h.vertices_visited - prof_v;
  r.edges = h.edges_visited - prof_e;
  r.bottom_up = h.mode;
  prof_v = h.vertices_visited;
  prof_e = h.edges_visited;
  // the group that only detected the empty frontier carries no work; a group that ran the last levels itself
  // (many levels per launch, grx_mid.hpp) and found the end is a record like any other
  if (h.done && r.frontier_size == 0 && r.edges == 0) ctx->levels.pop_back();
}

// After the last group: hints for the next search on the handle, errors raised on the device, run statistics.  The control
// block must be in ctx->h_ctrl (run_levels / grx_bfs_dist_end copied it, or the mailbox filled it: returned_fast).
grx_status_t bfs_search::finish(bool returned_fast, int groups_used, float* elapsed_ms) {
  hipStream_t s = ctx->stream;
  const bool hints = !part && !api_driven;
  if (hints && pace > 0 && groups_used > 0) {
    g->group_hint[dopt ? 1 : 0].store(groups_used, std::memory_order_relaxed);
    g->end_in_head[dopt ? 1 : 0].store(ctx->h_mailbox[12] == 0 ? 1 : 0, std::memory_order_relaxed);
  }
  if (hints && dopt && variant == 0) g->do_last_src.store((pace > 0 && groups_used > 0 && opt.max_iterations == 0) ? (uint32_t)src + 1u : 0u, std::memory_order_relaxed);
  if (launch_err != hipSuccess) { drop_events(); return fail(GRX_ERROR_HIP, hipGetErrorString(launch_err)); }
  if (ctx->h_mailbox[10] != 0 || (!returned_fast && ctx->h_ctrl->mid_err != 0)) {
    const int code = ctx->h_mailbox[10] != 0 ? (int)ctx->h_mailbox[10] : (int)ctx->h_ctrl->mid_err;
    ctx->h_mailbox[10] = 0;
    drop_events();
    if (code == 2) {
      // The sweep found that the bins did not receive exactly the level's out-edges: the per-XCD ticket queues of the
      // second scatter lost units (no workgroup of the launch ran on some XCD of the census).  Nothing wrong was
      // returned -- the search stopped there.  Repeat it with statically strided units, and keep that mode.
      GRX_HIP(hipStreamSynchronize(s));
      GRX_HIP(hipMemsetAsync(&ctx->d_ctrl->mid_err, 0, sizeof(int32_t), s));
      if (!ctx->sc2_static) {
        ctx->sc2_static = true;
        // (a rank of a partition cannot repeat its search alone: its caller gets the error and every later search of the
        // context draws its units statically)
        if (!part && !api_driven) return BFS_RETRY_STATIC;
      }
      return fail(GRX_ERROR_HIP, "grx_bfs: the binned scatter did not cover the level's edges (grx_bin.hpp)");
    }
    return fail(GRX_ERROR_HIP, "grx_bfs: a device-side barrier timed out (grx_mid.hpp)");
  }

  // OR-ed over the searches: a search from another source whose fat levels fall into other groups adds them
  // instead of replacing the set, so alternating sources do not keep evicting each other's groups.  A set bit costs
  // two no-op launches (~8 us) in a search that has no fat level there; GRX_BIN_HINT_REPLACE=1: the round-3 behaviour.
  if (hints && use_bins) {
    const uint32_t want = returned_fast ? (uint32_t)ctx->h_mailbox[11] : (uint32_t)ctx->h_ctrl->bin_want;
    if (env_int("GRX_BIN_HINT_REPLACE", 0) != 0) g->bin_hint.store(want, std::memory_order_relaxed);
    else g->bin_hint.fetch_or(want, std::memory_order_relaxed);
    if (!dopt && variant == 0 && opt.max_iterations == 0)
      g->bin_exact[profile ? 1 : 0].store(((uint64_t)((uint32_t)src + 1u) << 32) | (uint64_t)want, std::memory_order_relaxed);
  }
  float ms = 0;
  if (returned_fast) {
    // enact() time on the device's own clock: seed (init kernel) -> the kernel that found the
    // frontier empty; the reference brackets the same span with two events (enactor.hxx:270-282)
    ms = (float)((double)ctx->mailbox_ticks / ctx->wall_clock_khz);
  } else {
    GRX_HIP(hipEventRecord(ctx->ev_end, s));
    GRX_HIP(hipEventSynchronize(ctx->ev_end));
    GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  }
  drop_events();

  ctx->stats.edges_visited = ctx->h_ctrl->edges_visited;
  ctx->stats.vertices_visited = ctx->h_ctrl->vertices_visited;
  ctx->stats.search_depth = ctx->h_ctrl->level;
  ctx->stats.elapsed_ms = ms;
  ctx->stats.n_levels_recorded = (int32_t)ctx->levels.size();
  ctx->stats.reserved = variant == 3 ? (float)ctx->h_ctrl->spare[0] : (float)launches;
  if (elapsed_ms) *elapsed_ms = ms;
  return GRX_SUCCESS;
}

// the whole search of one rank that needs no exchange: setup, the device-driven level loop (run_levels), finish
static grx_status_t bfs_run_single(grx_context_t ctx, grx_graph_t g, int32_t src, const grx_options_t* options, int32_t* d_dist,
                                   float* elapsed_ms) {
  for (int attempt = 0; attempt < 2; ++attempt) {
    bfs_search S;
    grx_status_t st = S.setup(ctx, g, src, options, d_dist, nullptr, false, elapsed_ms);
    if (st != GRX_SUCCESS || S.answered) return st;
    bool returned_fast = false;
    int groups_used = 0;
    st = run_levels(ctx, S.opt, [&](hipStream_t stream, int seq) { S.launch_group(stream, seq); },
                    [&](const ctrl_t& h) { S.after_sync(h); }, /*first_batch=*/S.dense ? 8 : 4, /*pace_depth=*/S.pace,
                    /*fast_return=*/S.pace > 0 && ((S.opt.engine_flags & GRX_FLAG_ASYNC_RETURN) != 0 || env_int("GRX_FAST_RETURN", 0) != 0),
                    &returned_fast, S.hold_after, &groups_used);
    if (st != GRX_SUCCESS) { S.drop_events(); return st; }
    st = S.finish(returned_fast, groups_used, elapsed_ms);
    if (st != BFS_RETRY_STATIC) return st;
  }
  return fail(GRX_ERROR_HIP, "grx_bfs: the binned scatter did not cover the level's edges (grx_bin.hpp)");
}

extern "C" grx_status_t grx_bfs(grx_context_t ctx, grx_graph_t g, int32_t src,
                                const grx_options_t* options, int32_t* d_dist,
                                int32_t* d_pred, float* elapsed_ms) {
  (void)d_pred;  // accepted and never written, like the reference (bfs.hxx:29)
  if (!ctx || !g || !d_dist) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs: null argument");
  if (src < 0 || src >= g->V) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs: source out of range");
  env_scan_guard env_guard;
  return bfs_run_single(ctx, g, src, options, d_dist, elapsed_ms);
}

// ===============================================================================================================
// PARTITIONED (multi-GPU) BFS: the C ABI around bfs_search (include/grx.h, "multi-GPU").
//
// The reference is single-GPU only: every operator throws when `context.size() != 1`
// (advance/advance.hxx:129-132, filter/filter.hxx:96-99) and no NCCL/RCCL/MPI call exists in its tree.  This is the
// MI355X design:
//   * one process per GPU; rank r owns the vertex slice [r * S, min((r + 1) * S, V)), S a multiple of 2048, the OUT-rows of
//     that slice (global column ids) and -- for the bottom-up step -- its IN-rows (the same rows when the graph is symmetric);
//   * a rank runs the single-GPU ENGINE on its rows (round 6: the binned scatter + sweep pair on fat levels, the second
//     bottom-up body, the claim-per-edge advance on thin ones, chosen per rank and level by its own head kernel); at one rank
//     the partitioned search IS the single-GPU search, same object, same kernels, same launch schedule;
//   * every level moves exactly one fixed-size message per pair of GPUs: an S-bit bitmap.
//       top-down level : bit v of the slice sent to owner(v) = "I discovered v" (each vertex reported at most once per
//                        rank); the owner ORs the P - 1 slices it receives, claims the new ones and emits them as tiles;
//       bottom-up level: every rank sends its frontier slice to everybody (the all-to-all degenerates to an all-gather),
//                        so each rank holds the whole-graph frontier bitmap and scans the in-edges of its open vertices.
//     Fixed sizes mean NO size exchange and no host round trip: the host enqueues level groups blindly (kernels + the
//     bitmap all-to-all + a 4-word all-reduce of the frontier statistics) and reads `done` once per batch.  A slice is
//     S / 8 bytes (0.33 MB at 8 ranks on the 21 M-vertex stand-in): ~2 us on one 153 GB/s xGMI link, and on the full mesh
//     every pair has its own link;
//   * the end of the search and the direction (Beamer) are decided ON THE DEVICE from the all-reduced statistics, so all
//     ranks take the same branch without talking to the host; which BODY runs a level is each rank's own choice.
// ===============================================================================================================
#include <rccl/rccl.h>  // types only: the library is opened at run time (rccl_api below)

#include <dlfcn.h>

// RCCL, resolved at run time: libgrx.so carries no link-time dependency on it (single-GPU users
// never load it), and the collectives of a level group can be issued from C -- one host call per
// group, capturable into a HIP graph without any Python in between.
struct rccl_api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
static rccl_api& rccl() {
  static rccl_api api = [] {
    rccl_api a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (!a.lib) return a;
    auto sym = [&](const char* n) { return dlsym(a.lib, n); };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv &&
           a.AllReduce && a.GetErrorString;
    return a;
  }();
  return api;
}
#define GRX_NCCL(expr)                                                                         \
  do {                                                                                         \
    ncclResult_t _r = (expr);                                                                  \
    if (_r != ncclSuccess)                                                                     \
      return ::grx::fail(GRX_ERROR_HIP, std::string("RCCL: ") + rccl().GetErrorString(_r) + "\t: " #expr); \
  } while (0)

constexpr int DIST_MAX_RANKS = 64;

struct grx_bfs_dist {
  ncclComm_t comm = nullptr;             // own communicator (grx_bfs_dist_comm_init), null: the caller runs the collectives
  hipGraphExec_t group_graph = nullptr;  // one captured level group (kernels + both collectives)
  int32_t* graph_labels = nullptr;       // label base pointer / direction the captured group was recorded for
  int32_t graph_dir = -1;
  bool graph_failed = false;
  grx_context_t ctx = nullptr;
  grx_graph_t g = nullptr;      // out-rows of the owned slice
  bfs_part_cfg pc{};
  int32_t parts = 1;
  bfs_search S{};               // the search in flight
  int32_t seq = 0;              // level groups enqueued by grx_bfs_dist_pre since begin
  int32_t last_groups = 0;      // groups the previous whole-search call needed (grx_bfs_dist_run: its first batch)
  bool active = false;
};

extern "C" {

int32_t grx_bfs_dist_slice_bits(int32_t n_vertices, int32_t n_ranks) {
  if (n_vertices < 0 || n_ranks < 1) return 0;
  const long long per = ((long long)n_vertices + n_ranks - 1) / n_ranks;
  const long long s = ((per + 2047) / 2048) * 2048;
  return (int32_t)(s < 2048 ? 2048 : s);
}

grx_status_t grx_bfs_dist_create(grx_context_t ctx, grx_graph_t out_rows, grx_graph_t in_rows, int32_t n_ranks,
                                 int32_t my_rank, long long n_edges_global, int32_t parts, void* d_send,
                                 void* d_recv, long long* d_stats_local, const long long* d_stats_global,
                                 grx_bfs_dist_t* out) {
  if (!ctx || !out_rows || !d_send || !d_recv || !d_stats_local || !d_stats_global || !out)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_create: null argument");
  if (n_ranks < 1 || n_ranks > DIST_MAX_RANKS || my_rank < 0 || my_rank >= n_ranks)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_create: bad rank layout");
  if (parts != 1 && parts != 2) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_create: parts must be 1 or 2");
  if (in_rows && in_rows->V != out_rows->V)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_create: in-rows and out-rows disagree on V");
  GRX_HIP(hipSetDevice(ctx->device));
  grx_bfs_dist* h = new grx_bfs_dist();
  h->ctx = ctx;
  h->g = out_rows;
  h->parts = parts;
  const int32_t S = grx_bfs_dist_slice_bits(out_rows->V, n_ranks);
  const long long lo = (long long)my_rank * S, hi = lo + S;
  part_args& x = h->pc.x;
  x.P = n_ranks;
  x.rank = my_rank;
  x.lo = (int32_t)(lo < out_rows->V ? lo : out_rows->V);
  x.hi = (int32_t)(hi < out_rows->V ? hi : out_rows->V);
  x.slice_words = S / 32;
  x.send = static_cast<unsigned*>(d_send);   // half 0 of [parts][n_ranks][slice_words] (see grx_bfs_dist_pre)
  x.recv = static_cast<const unsigned*>(d_recv);
  x.stats_local = d_stats_local;
  x.stats_global = d_stats_global;
  x.e_global = n_edges_global;
  h->pc.g_in = in_rows;
  *out = h;
  return GRX_SUCCESS;
}

// problem.reset() + frontier seed.  The caller all-reduces stats_local into stats_global
// before the first grx_bfs_dist_pre.  advance_direction: GRX_DIR_FORWARD keeps every level
// top-down; GRX_DIR_OPTIMIZED enables the bottom-up step (needs in-rows or symmetry).
// Sharded labels: d_local holds ONLY the owned slice (S = grx_bfs_dist_slice_bits entries, vertex v at
// d_local[v - rank * S]).  Every kernel of a partitioned search dereferences labels of owned vertices only,
// so they all work on the base pointer d_local - lo.
grx_status_t grx_bfs_dist_begin_local(grx_bfs_dist_t h, int32_t source, int32_t advance_direction, int32_t* d_local) {
  if (!h || !d_local) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin_local: null argument");
  return grx_bfs_dist_begin(h, source, advance_direction, d_local - h->pc.x.lo);
}

grx_status_t grx_bfs_dist_begin(grx_bfs_dist_t h, int32_t source, int32_t advance_direction, int32_t* d_dist) {
  if (!h || !d_dist) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin: null argument");
  if (source < 0 || source >= h->g->V) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin: source out of range");
  env_scan_guard env_guard;
  grx_options_t opt;
  grx_options_default(&opt);
  opt.advance_direction = advance_direction;
  h->S = bfs_search{};
  h->seq = 0;
  grx_status_t st = h->S.setup(h->ctx, h->g, source, &opt, d_dist, &h->pc, /*api_driven=*/true, nullptr);
  if (st != GRX_SUCCESS) return st;
  if (h->pc.x.P == 1) {
    // one rank: the statistics words play no part (the heads look at the rank's own frontier), but the protocol reads them
    GRX_HIP(hipMemsetAsync(h->pc.x.stats_local, 0, 4 * sizeof(long long), h->ctx->stream));
  }
  h->active = true;
  return GRX_SUCCESS;
}

// Enqueue the part of a level group that precedes the exchange: head, level kernel(s) -- part 0.  After it the caller
// exchanges send -> recv (all_to_all_single, n_ranks equal splits of slice_words words).  Asynchronous.
// parts == 2 (kept for callers written against the two-halves protocol): half 1 of the buffers carries nothing -- every
// report of a level travels in half 0 -- and part 1 is a no-op.
grx_status_t grx_bfs_dist_pre(grx_bfs_dist_t h, int32_t part) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_pre: no BFS in flight");
  if (part < 0 || part >= h->parts) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_pre: bad part");
  if (part != 0) return GRX_SUCCESS;
  env_scan_guard env_guard;
  // (group indices from 32 on: a generic group -- every kernel a level may need, no launch-schedule hints)
  h->S.launch_group(h->ctx->stream, 32 + h->seq++);
  if (h->S.launch_err != hipSuccess) return fail(GRX_ERROR_HIP, hipGetErrorString(h->S.launch_err));
  return GRX_SUCCESS;
}

// Enqueue the part of a level group that follows the exchange: bottom-up level / claim of the reports, then
// the statistics kernel.  The caller then all-reduces stats_local into stats_global.
grx_status_t grx_bfs_dist_post(grx_bfs_dist_t h) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_post: no BFS in flight");
  h->S.launch_post(h->ctx->stream);
  if (h->S.launch_err != hipSuccess) return fail(GRX_ERROR_HIP, hipGetErrorString(h->S.launch_err));
  return GRX_SUCCESS;
}

// Wait for everything enqueued so far and report the state of the search.
grx_status_t grx_bfs_dist_poll(grx_bfs_dist_t h, int32_t* done, int32_t* level) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_poll: no BFS in flight");
  grx_context_t ctx = h->ctx;
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, ctx->stream));
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  if (done) *done = ctx->h_ctrl->done;
  if (level) *level = ctx->h_ctrl->level;
  return GRX_SUCCESS;
}

grx_status_t grx_bfs_dist_end(grx_bfs_dist_t h, grx_run_stats_t* stats) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_end: no BFS in flight");
  grx_context_t ctx = h->ctx;
  env_scan_guard env_guard;
  h->active = false;
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, ctx->stream));
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  float ms = 0;
  grx_status_t st = h->S.finish(false, 0, &ms);
  if (st != GRX_SUCCESS) return st;
  ctx->stats.n_levels_recorded = 0;
  if (stats) *stats = ctx->stats;
  return GRX_SUCCESS;
}

// ---- RCCL transport inside the library --------------------------------------------------------
int32_t grx_dist_unique_id_bytes(void) { return (int32_t)sizeof(ncclUniqueId); }

grx_status_t grx_dist_unique_id(void* out) {
  if (!out) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_dist_unique_id: null argument");
  if (!rccl().ok) return fail(GRX_ERROR_UNSUPPORTED, "grx_dist_unique_id: librccl could not be opened");
  GRX_NCCL(rccl().GetUniqueId(reinterpret_cast<ncclUniqueId*>(out)));
  return GRX_SUCCESS;
}

grx_status_t grx_bfs_dist_comm_init(grx_bfs_dist_t h, const void* unique_id) {
  if (!h || !unique_id) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_comm_init: null argument");
  if (!rccl().ok) return fail(GRX_ERROR_UNSUPPORTED, "grx_bfs_dist_comm_init: librccl could not be opened");
  GRX_HIP(hipSetDevice(h->ctx->device));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  GRX_NCCL(rccl().CommInitRank(&h->comm, h->pc.x.P, id, h->pc.x.rank));
  return GRX_SUCCESS;
}

// the two collectives of a level group on the context's stream: the bitmap all-to-all (grouped send / recv: every pair of
// GPUs has its own xGMI link) and the 4-word statistics all-reduce
static grx_status_t dist_exchange(grx_bfs_dist_t h) {
  const part_args& x = h->pc.x;
  hipStream_t s = h->ctx->stream;
  GRX_NCCL(rccl().GroupStart());
  for (int j = 0; j < x.P; ++j) {
    GRX_NCCL(rccl().Send(x.send + (size_t)j * x.slice_words, (size_t)x.slice_words, ncclUint32, j, h->comm, s));
    GRX_NCCL(rccl().Recv(const_cast<unsigned*>(x.recv) + (size_t)j * x.slice_words, (size_t)x.slice_words, ncclUint32, j,
                         h->comm, s));
  }
  GRX_NCCL(rccl().GroupEnd());
  return GRX_SUCCESS;
}
static grx_status_t dist_allreduce(grx_bfs_dist_t h) {
  const part_args& x = h->pc.x;
  GRX_NCCL(rccl().AllReduce(x.stats_local, const_cast<long long*>(x.stats_global), 4, ncclInt64, ncclSum, h->comm, h->ctx->stream));
  return GRX_SUCCESS;
}
// one level group, everything on the context's stream
static grx_status_t dist_group_enqueue(grx_bfs_dist_t h) {
  grx_status_t st = grx_bfs_dist_pre(h, 0);
  if (st != GRX_SUCCESS) return st;
  if ((st = dist_exchange(h)) != GRX_SUCCESS) return st;
  if ((st = grx_bfs_dist_post(h)) != GRX_SUCCESS) return st;
  return dist_allreduce(h);
}

// the all-reduce that follows grx_bfs_dist_begin (frontier statistics of the seed)
grx_status_t grx_bfs_dist_seed_stats(grx_bfs_dist_t h) {
  if (!h || !h->comm) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_seed_stats: no communicator");
  return dist_allreduce(h);
}

// Enqueue n level groups.  A group of a PARTITION takes no level-dependent argument, so once a search has finished the
// group is captured into a HIP graph (grx_bfs_dist_capture_group) and later calls replay it: one graph
// launch per level; without a captured graph the groups are enqueued eagerly.
grx_status_t grx_bfs_dist_groups(grx_bfs_dist_t h, int32_t n) {
  if (!h || !h->active || !h->comm) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_groups: no BFS in flight or no communicator");
  const bool replay = h->group_graph && h->graph_labels == h->S.d_dist && h->graph_dir == (h->S.dopt ? 1 : 0);
  for (int i = 0; i < n; ++i) {
    if (replay) {
      GRX_HIP(hipGraphLaunch(h->group_graph, h->ctx->stream));
    } else {
      grx_status_t st = dist_group_enqueue(h);
      if (st != GRX_SUCCESS) return st;
    }
  }
  return GRX_SUCCESS;
}

// Record one level group for the label buffer / direction of the search that just ended (every kernel of a
// further group exits on `done`, and capture only records).  Must be called on every rank alike.  A failure
// is remembered and leaves the eager path in place.
grx_status_t grx_bfs_dist_capture_group(grx_bfs_dist_t h) {
  if (!h || !h->comm) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_capture_group: no communicator");
  if (h->graph_failed || !h->S.ctx) return GRX_SUCCESS;
  const int dir = h->S.dopt ? 1 : 0;
  if (h->group_graph && h->graph_labels == h->S.d_dist && h->graph_dir == dir) return GRX_SUCCESS;
  if (h->group_graph) { (void)hipGraphExecDestroy(h->group_graph); h->group_graph = nullptr; }
  hipStream_t s = h->ctx->stream;
  hipGraph_t graph = nullptr;
  const bool was_active = h->active;
  h->active = true;
  bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
  if (ok) {
    ok = dist_group_enqueue(h) == GRX_SUCCESS;
    ok = (hipStreamEndCapture(s, &graph) == hipSuccess) && ok && graph != nullptr;
    if (ok) ok = hipGraphInstantiate(&h->group_graph, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
  }
  (void)hipGetLastError();
  h->S.launch_err = hipSuccess;
  h->active = was_active;
  if (!ok) {
    h->group_graph = nullptr;
    h->graph_failed = true;
    return GRX_SUCCESS;
  }
  h->graph_labels = h->S.d_dist;
  h->graph_dir = dir;
  return GRX_SUCCESS;
}

int32_t grx_bfs_dist_group_is_captured(grx_bfs_dist_t h) { return (h && h->group_graph) ? 1 : 0; }

// A WHOLE search in one call (round 6).  One rank: exactly grx_bfs -- the same object, kernels and paced launch schedule
// (bfs_run_single); the partition of one slice is the graph.  More ranks: needs the in-library transport
// (grx_bfs_dist_comm_init): reset + seed, the seed's all-reduce, then batches of level groups -- the first as long as the
// previous search on the handle (every rank sees the same depth, so every rank enqueues the same number of collectives) --
// with one look at `done` per batch.  labels_are_local != 0: d_labels is the owned slice (see grx_bfs_dist_begin_local).
grx_status_t grx_bfs_dist_run(grx_bfs_dist_t h, int32_t source, int32_t advance_direction, int32_t* d_labels,
                              int32_t labels_are_local, grx_run_stats_t* stats) {
  if (!h || !d_labels) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_run: null argument");
  if (h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_run: a search driven by level groups is in flight");
  if (source < 0 || source >= h->g->V) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_run: source out of range");
  int32_t* d_dist = labels_are_local ? d_labels - h->pc.x.lo : d_labels;
  if (h->pc.x.P == 1) {
    env_scan_guard env_guard;
    grx_options_t opt;
    grx_options_default(&opt);
    opt.advance_direction = advance_direction;
    opt.engine_flags = GRX_FLAG_NO_BLOCK_ASYNC;
    float ms = 0;
    grx_status_t st = bfs_run_single(h->ctx, h->g, source, &opt, d_dist, &ms);
    if (st != GRX_SUCCESS) return st;
    if (stats) *stats = h->ctx->stats;
    return GRX_SUCCESS;
  }
  if (!h->comm) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_run: more than one rank needs grx_bfs_dist_comm_init (or the level-group calls)");
  grx_status_t st = grx_bfs_dist_begin(h, source, advance_direction, d_dist);
  if (st != GRX_SUCCESS) return st;
  if ((st = dist_allreduce(h)) != GRX_SUCCESS) return st;
  int batch = h->last_groups > 0 ? std::max(2, std::min(h->last_groups, 64)) : 4;
  for (;;) {
    if ((st = grx_bfs_dist_groups(h, batch)) != GRX_SUCCESS) return st;
    int32_t done = 0;
    if ((st = grx_bfs_dist_poll(h, &done, nullptr)) != GRX_SUCCESS) return st;
    if (done) break;
    batch = std::min(batch * 2, 32);
  }
  (void)grx_bfs_dist_capture_group(h);  // no-op once recorded for this buffer / direction
  h->active = true;
  st = grx_bfs_dist_end(h, stats);
  if (st == GRX_SUCCESS) h->last_groups = h->ctx->stats.search_depth + 1;  // + the group whose head finds the frontier empty
  return st;
}

grx_status_t grx_bfs_dist_destroy(grx_bfs_dist_t h) {
  if (h) {
    if (h->group_graph) (void)hipGraphExecDestroy(h->group_graph);
    if (h->comm && rccl().ok) (void)rccl().CommDestroy(h->comm);
  }
  delete h;
  return GRX_SUCCESS;
}

}  // extern "C"
