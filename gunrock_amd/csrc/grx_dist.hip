// grx_dist.hip -- level-stepping interface of the BFS enactor for multi-GPU hosts.
//
// The reference is single-GPU only: every operator throws when
// `context.size() != 1` (advance/advance.hxx:129-132, filter/filter.hxx:96-99) and no
// NCCL/RCCL/MPI call exists in its tree.  This file is the device side of the
// MI355X multi-GPU design:
//   * 1-D vertex-range partition: rank r owns vertices [bounds[r], bounds[r+1]) and the
//     CSR rows of those vertices (GLOBAL column ids; rows of other ranks are empty), plus
//     a full-size label array in which only the owned range is authoritative;
//   * per level: advance (the same fused kernel as single-GPU) -> COLLECT the winners
//     this rank does not own into one bucket per owner -> the host exchanges buckets with
//     an RCCL all-to-all (one process per GPU, torch.distributed) -> APPLY the received
//     candidates (claim with atomicMin, append winners to the next frontier);
//   * termination: all-reduce of the next frontier sizes (host).
// A rank marks every vertex it has ever sent in its own label copy, so a vertex crosses
// a given link at most once per BFS.
#include "grx_engine.hpp"

#include <climits>

namespace grx {

constexpr int DIST_MAX_RANKS = 64;

struct dist_args {
  int32_t bounds[DIST_MAX_RANKS + 1];
  int32_t n_ranks;
  int32_t my_rank;
  int32_t* send;             // bucket of owner j starts at send + bounds[j] (capacity = its vertex count)
  unsigned long long* cursor;  // [n_ranks], 16 words apart (own cache line each)
  int32_t* dist;
};

constexpr int CURSOR_STRIDE = 16;

// Same claim as single-GPU top-down BFS (grx_bfs.hip, variant 0).
struct bfs_policy_dist {
  using src_state = int;
  int32_t* dist;
  int next_depth;
  __device__ __forceinline__ void begin(ctrl_t* c) { next_depth = c->level + 1; }
  __device__ __forceinline__ src_state load_source(int) const { return 0; }
  __device__ __forceinline__ bool precheck(src_state, int n, int) const { return dist[n] > next_depth; }
  __device__ __forceinline__ bool visit(int, src_state, int n, int) const {
    return next_depth < atomicMin(&dist[n], next_depth);
  }
};

__global__ void dist_counts_kernel(const unsigned long long* cursor, int n_ranks, long long* out) {
  if ((int)threadIdx.x < n_ranks) out[threadIdx.x] = (long long)cursor[threadIdx.x * CURSOR_STRIDE];
}

// Start of a level: level counter, statistics of the frontier entering it (owned
// vertices only), reset of the output side.  Never sets `done`: another rank may
// still feed this one.  <<<1, 1024>>>
__global__ __launch_bounds__(PLAN_BLOCK) void dist_level_begin_kernel(pipe_args a, dist_args d) {
  __shared__ unsigned long long s_n, s_m;
  ctrl_t* c = a.ctrl;
  const int tid = threadIdx.x;
  const int level = c->level + 1;
  const int p = level & 1;
  const int nt = c->n_tiles[p];
  if (tid == 0) { s_n = 0; s_m = 0; }
  __syncthreads();
  long long n = 0, m = 0;
  for (int i = tid; i < nt; i += PLAN_BLOCK) {
    n += a.tile_count[i];
    m += a.tile_sums[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_xor(n, o, 64);
    m += __shfl_xor(m, o, 64);
  }
  if (dev::lane_id() == 0) {
    atomicAdd(&s_n, (unsigned long long)n);
    atomicAdd(&s_m, (unsigned long long)m);
  }
  __syncthreads();
  if (tid < d.n_ranks) d.cursor[tid * CURSOR_STRIDE] = 0ull;
  if (tid != 0) return;
  c->done = 0;
  c->mode = 0;
  c->level = level;
  c->edges_visited += (long long)s_m;
  c->vertices_visited += (long long)s_n;
  c->n_items[p] = (int)s_n;
  c->q_edges[p] = (long long)s_m;
  c->n_tiles[p ^ 1] = 0;
}

__device__ __forceinline__ int owner_of(const dist_args& d, int v) {
  int o = 0;
  for (int j = 1; j < d.n_ranks; ++j) o += (v >= d.bounds[j]) ? 1 : 0;
  return o;
}

// Move the winners this rank does not own out of the freshly produced frontier
// (parity q = (level + 1) & 1) into per-owner buckets.  Two sweeps per workgroup over
// its tiles: count (LDS histogram) -> one cursor atomic per owner per workgroup ->
// write.  Removed slots become -1 and the tile's vertex count is corrected (their
// local degree is 0, so degree sums and chunk counts are unaffected).
__global__ __launch_bounds__(ADV_BLOCK) void dist_collect_kernel(pipe_args a, dist_args d) {
  __shared__ int s_hist[DIST_MAX_RANKS];
  __shared__ int s_base[DIST_MAX_RANKS];
  __shared__ int s_removed;
  ctrl_t* c = a.ctrl;
  const int q = (c->level + 1) & 1;
  const int nt = c->n_tiles[q];
  const int tid = threadIdx.x;
  int32_t* out = a.frontier[q];
  const int lo = d.bounds[d.my_rank], hi = d.bounds[d.my_rank + 1];
  if (tid < DIST_MAX_RANKS) s_hist[tid] = 0;
  __syncthreads();
  for (int t = blockIdx.x; t < nt; t += gridDim.x) {
    if (a.tile_count[t] == 0) continue;  // reserved-but-unused tile: its slots are stale
    const int v = out[(size_t)t * TILE + tid];
    if (v >= 0 && (v < lo || v >= hi)) atomicAdd(&s_hist[owner_of(d, v)], 1);
  }
  __syncthreads();
  if (tid < d.n_ranks) {
    const int cnt = s_hist[tid];
    s_base[tid] = cnt ? (int)atomicAdd(&d.cursor[tid * CURSOR_STRIDE], (unsigned long long)cnt) : 0;
    s_hist[tid] = 0;  // becomes the running offset inside this workgroup's share
  }
  __syncthreads();
  for (int t = blockIdx.x; t < nt; t += gridDim.x) {
    if (a.tile_count[t] == 0) continue;
    if (tid == 0) s_removed = 0;
    __syncthreads();
    const size_t slot = (size_t)t * TILE + tid;
    const int v = out[slot];
    if (v >= 0 && (v < lo || v >= hi)) {
      const int o = owner_of(d, v);
      const int at = s_base[o] + atomicAdd(&s_hist[o], 1);
      d.send[d.bounds[o] + at] = v;
      out[slot] = -1;
      atomicAdd(&s_removed, 1);
    }
    __syncthreads();
    if (tid == 0 && s_removed) a.tile_count[t] -= s_removed;
    __syncthreads();
  }
}

// Claim the candidates received from the other ranks (all owned by this rank) and
// append the winners to the next frontier as tiles.
__global__ __launch_bounds__(ADV_BLOCK) void dist_apply_kernel(pipe_args a, dist_args d, const int32_t* recv,
                                                               long long n) {
  __shared__ int s_out[2 * TILE];
  __shared__ int s_wave[ADV_BLOCK / 64 + 1];
  __shared__ int s_res[3];
  __shared__ int s_cnt;
  ctrl_t* c = a.ctrl;
  const int depth = c->level + 1;
  const int q = depth & 1;
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  if (tid == 0) { s_cnt = 0; s_res[0] = 0; s_res[1] = 0; }
  __syncthreads();
  for (long long base = (long long)blockIdx.x * ADV_BLOCK; base < n; base += (long long)gridDim.x * ADV_BLOCK) {
    const long long i = base + tid;
    bool win = false;
    int v = -1;
    if (i < n) {
      v = recv[i];
      if (d.dist[v] > depth) win = depth < atomicMin(&d.dist[v], depth);
    }
    const unsigned long long m = dev::ballot(win);
    if (m) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&s_cnt, __popcll(m));
      at = __shfl(at, 0, 64);
      if (win) s_out[at + dev::mask_rank(m)] = v;
    }
    __syncthreads();
    int have = s_cnt;
    __syncthreads();
    if (have >= TILE) {
      emit_tile(a, c, q, s_out, have - TILE, TILE, s_wave, s_res);
      have -= TILE;
      __syncthreads();
    }
    if (tid == 0) s_cnt = have;
    __syncthreads();
  }
  const int rem = s_cnt;
  if (rem > 0) emit_tile(a, c, q, s_out, 0, rem, s_wave, s_res);
  __syncthreads();
  release_tiles(a, s_res);
}

// Size of the frontier the next level will expand.  <<<1, 1024>>>
__global__ __launch_bounds__(PLAN_BLOCK) void dist_frontier_size_kernel(pipe_args a, long long* out) {
  __shared__ unsigned long long s_n, s_m;
  ctrl_t* c = a.ctrl;
  const int q = (c->level + 1) & 1;
  const int nt = c->n_tiles[q];
  if (threadIdx.x == 0) { s_n = 0; s_m = 0; }
  __syncthreads();
  long long n = 0, m = 0;
  for (int i = threadIdx.x; i < nt; i += PLAN_BLOCK) {
    n += a.tile_count[i];
    m += a.tile_sums[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_xor(n, o, 64);
    m += __shfl_xor(m, o, 64);
  }
  if (dev::lane_id() == 0) {
    atomicAdd(&s_n, (unsigned long long)n);
    atomicAdd(&s_m, (unsigned long long)m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = (long long)s_n;
    out[1] = (long long)s_m;
  }
}

__global__ void dist_init_kernel(pipe_args a, int32_t* dist, int src) {
  const int tid = threadIdx.x;
  a.frontier[0][tid] = (tid == 0 && src >= 0) ? src : -1;
  if (tid == 0) {
    ctrl_t* c = a.ctrl;
    const int deg = src >= 0 ? a.ro[src + 1] - a.ro[src] : 0;
    a.tile_sums[0] = deg;
    a.tile_chunks[0] = (deg + CHUNK - 1) / CHUNK;
    a.tile_count[0] = src >= 0 ? 1 : 0;
    c->level = -1;
    c->done = 0;
    c->mode = 0;
    c->n_tiles[0] = src >= 0 ? 1 : 0;
    c->n_tiles[1] = 0;
    c->n_items[0] = c->n_items[1] = 0;
    c->total_chunks = 0;
    c->edges_visited = 0;
    c->vertices_visited = 0;
    c->frontier_bitmap = 0;
    c->convert = 0;
    if (src >= 0) dist[src] = 0;
  }
}

struct dist_state {
  pipe_args a;
  dist_args d;
  bool active = false;
};

static dist_state& state_of(grx_context_t ctx) {
  static thread_local dist_state st;  // one BFS in flight per host thread
  (void)ctx;
  return st;
}

}  // namespace grx

using namespace grx;

extern "C" {

grx_status_t grx_bfs_dist_begin(grx_context_t ctx, grx_graph_t g, int32_t source_if_owned,
                                const int32_t* bounds, int32_t n_ranks, int32_t my_rank, int32_t* d_send,
                                int32_t* d_dist) {
  if (!ctx || !g || !bounds || !d_send || !d_dist)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin: null argument");
  if (n_ranks < 1 || n_ranks > DIST_MAX_RANKS || my_rank < 0 || my_rank >= n_ranks)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin: bad rank layout");
  if (bounds[0] != 0 || bounds[n_ranks] != g->V)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin: bounds must cover [0, V]");
  if (source_if_owned >= 0 && (source_if_owned < bounds[my_rank] || source_if_owned >= bounds[my_rank + 1]))
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin: source not owned by this rank");
  GRX_HIP(hipSetDevice(ctx->device));
  dist_state& st = state_of(ctx);
  grx_status_t rc = pipeline_prepare(ctx, g, &st.a);
  if (rc != GRX_SUCCESS) return rc;
  GRX_HIP(ctx->misc.reserve((size_t)DIST_MAX_RANKS * CURSOR_STRIDE * sizeof(unsigned long long) + 64));
  for (int i = 0; i <= n_ranks; ++i) st.d.bounds[i] = bounds[i];
  st.d.n_ranks = n_ranks;
  st.d.my_rank = my_rank;
  st.d.send = d_send;
  st.d.cursor = ctx->misc.as<unsigned long long>() + 8;
  st.d.dist = d_dist;
  hipStream_t s = ctx->stream;
  GRX_HIP(fill_i32(s, d_dist, INT_MAX, g->V));
  GRX_HIP(hipEventRecord(ctx->ev_begin, s));
  hipLaunchKernelGGL(dist_init_kernel, dim3(1), dim3(TILE), 0, s, st.a, d_dist, source_if_owned);
  GRX_HIP(hipGetLastError());
  st.active = true;
  return GRX_SUCCESS;
}

// One level: expand the owned frontier, then bin the non-owned winners by owner.
// d_counts[n_ranks] (device, int64) receives the bucket sizes.  Asynchronous.
grx_status_t grx_bfs_dist_advance(grx_context_t ctx, long long* d_counts) {
  dist_state& st = state_of(ctx);
  if (!st.active || !d_counts) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_advance: no BFS in flight");
  hipStream_t s = ctx->stream;
  const int grid = advance_grid(ctx);
  hipLaunchKernelGGL(dist_level_begin_kernel, dim3(1), dim3(PLAN_BLOCK), 0, s, st.a, st.d);
  hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(PLAN_BLOCK), 0, s, st.a, 1);
  hipLaunchKernelGGL((advance_kernel<bfs_policy_dist>), dim3(grid), dim3(ADV_BLOCK), 0, s, st.a,
                     bfs_policy_dist{st.d.dist, 0});
  hipLaunchKernelGGL(dist_collect_kernel, dim3(grid / 4), dim3(ADV_BLOCK), 0, s, st.a, st.d);
  hipLaunchKernelGGL(dist_counts_kernel, dim3(1), dim3(DIST_MAX_RANKS), 0, s, st.d.cursor, st.d.n_ranks, d_counts);
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

// Claim `n` received candidates (global ids owned by this rank).  Asynchronous.
grx_status_t grx_bfs_dist_apply(grx_context_t ctx, const int32_t* d_recv, long long n) {
  dist_state& st = state_of(ctx);
  if (!st.active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_apply: no BFS in flight");
  if (n <= 0) return GRX_SUCCESS;
  long long blocks = (n + ADV_BLOCK - 1) / ADV_BLOCK;
  const long long cap = advance_grid(ctx) / 2;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(dist_apply_kernel, dim3((unsigned)blocks), dim3(ADV_BLOCK), 0, ctx->stream, st.a, st.d, d_recv, n);
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

// {vertices, out-edges} of the frontier the next level will expand.  Synchronises.
grx_status_t grx_bfs_dist_frontier(grx_context_t ctx, long long* n_vertices, long long* n_edges) {
  dist_state& st = state_of(ctx);
  if (!st.active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_frontier: no BFS in flight");
  long long* d_out = reinterpret_cast<long long*>(ctx->misc.as<unsigned long long>());
  hipLaunchKernelGGL(dist_frontier_size_kernel, dim3(1), dim3(PLAN_BLOCK), 0, ctx->stream, st.a, d_out);
  long long h[2] = {0, 0};
  GRX_HIP(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  if (n_vertices) *n_vertices = h[0];
  if (n_edges) *n_edges = h[1];
  return GRX_SUCCESS;
}

grx_status_t grx_bfs_dist_end(grx_context_t ctx, grx_run_stats_t* stats) {
  dist_state& st = state_of(ctx);
  if (!st.active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_end: no BFS in flight");
  hipStream_t s = ctx->stream;
  GRX_HIP(hipEventRecord(ctx->ev_end, s));
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, s));
  GRX_HIP(hipEventSynchronize(ctx->ev_end));
  GRX_HIP(hipStreamSynchronize(s));
  float ms = 0;
  GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  ctx->stats.edges_visited = ctx->h_ctrl->edges_visited;
  ctx->stats.vertices_visited = ctx->h_ctrl->vertices_visited;
  ctx->stats.search_depth = ctx->h_ctrl->level + 1;
  ctx->stats.elapsed_ms = ms;
  ctx->stats.n_levels_recorded = 0;
  if (stats) *stats = ctx->stats;
  st.active = false;
  return GRX_SUCCESS;
}

}  // extern "C"
