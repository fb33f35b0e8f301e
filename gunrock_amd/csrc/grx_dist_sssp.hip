// grx_dist_sssp.hip -- partitioned (multi-GPU) SSSP: device side + level-group C ABI.
//
// The reference is single-GPU only (every operator throws when context.size() != 1:
// advance/advance.hxx:129-132); its SSSP is the frontier relaxation of algorithms/sssp.hxx:104-159
// (advance with atomicMin on the tentative distance, filter on "visited this iteration").  This is
// the same recurrence on a 1-D vertex partition, in the style of the partitioned BFS
// (grx_dist.hip): one process per GPU, rank r owns the vertex slice [r * S, min((r + 1) * S, V)) and
// the out-rows of that slice (global column ids); labels are SHARDED.
//
//   * every iteration moves ONE FIXED-SIZE message per pair of GPUs: S floats -- entry v - lo_j of the
//     message to rank j = the smallest tentative distance this rank found for j's vertex v in this
//     iteration (FLT_MAX: none).  A relaxation whose target is owned is claimed on the label exactly
//     as on one GPU; a remote target is min-reduced into the outgoing array instead (same atomic,
//     other address).  After the all-to-all the owner takes the minimum of the P - 1 slices it
//     received, lowers its label where that improves it and appends the vertex to its next frontier
//     (de-duplicated against its own discoveries of the iteration through the iteration stamps).
//   * fixed sizes mean no size exchange and no host round trip: the host enqueues iteration groups
//     blindly (kernels + all_to_all_single + an all_reduce of the next frontier's size) and reads
//     `done` once per batch; groups after `done` are no-ops on every rank alike.
//   * volume: 4 S bytes per peer and iteration whatever the frontier -- right for the low-diameter
//     graphs a partitioned search is for (tens of iterations); a road network's thousands of
//     near-empty iterations would be dominated by it (the single-GPU engine's delta-stepping
//     schedule is not partitioned).
// Distances equal the single-GPU engine's and the CPU oracle's bit for bit: the fixed point of
// d[v] = min(d[v], fl(d[u] + w(u, v))) does not depend on the schedule.
#include "grx_engine.hpp"
#include "grx_bfs_kernels.hpp"

#include <cfloat>

namespace grx {

struct sdist_args {
  int32_t n_ranks, my_rank;
  int32_t lo, hi;          // owned vertices
  int32_t S;               // slice size (vertices per rank)
  float* send;             // [n_ranks][S]: slice j is sent to rank j; addressed by GLOBAL vertex id (slices are S apart)
  const float* recv;       // [n_ranks][S]: slice j = what rank j found for the vertices of this rank
  long long* stats_local;  // [0] frontier vertices of the next iteration (this rank), [1] their out-edges
  const long long* stats_global;
};

struct sssp_policy_dist {
  using src_state = float;
  float* dist;        // sharded labels through a base pointer: only owned ids are dereferenced
  int32_t* stamp;     // the same for the iteration stamps
  const float* w;     // null: 1.0
  float* send;
  int lo, hi;
  int level;
  __device__ __forceinline__ void begin(ctrl_t* c) { level = c->level; }
  __device__ __forceinline__ bool owned(int n) const { return n >= lo && n < hi; }
  __device__ __forceinline__ src_state load_source(int v) const { return dist[v]; }
  __device__ __forceinline__ float edge_weight(int e) const { return w ? w[e] : 1.0f; }
  static constexpr bool two_claims = true;
  // ONE load from a selected address: the label of an owned target, the pending minimum of a remote one
  __device__ __forceinline__ bool precheck(src_state d_src, int n, int e, int& cand) const {
    const float nd = d_src + edge_weight(e);
    cand = __float_as_int(nd);
    const float* p = owned(n) ? dist + n : send + n;
    return nd < *p;
  }
  __device__ __forceinline__ int claim(int n, int cand) const {
    float* p = owned(n) ? dist + n : send + n;
    return __float_as_int(dev::atomic_min_f32(p, __int_as_float(cand)));
  }
  __device__ __forceinline__ bool improved(int raw1, int cand) const { return __int_as_float(cand) < __int_as_float(raw1); }
  __device__ __forceinline__ bool need2(int raw1, int cand) const { return improved(raw1, cand); }
  // once per iteration and owned vertex; a remote target never joins this rank's frontier
  __device__ __forceinline__ int claim2(int n) const { return owned(n) ? atomicExch(&stamp[n], level) : level; }
  __device__ __forceinline__ int code(int raw1, int raw2, int, int cand) const {
    return (improved(raw1, cand) && raw2 != level) ? 1 : 0;
  }
  __device__ __forceinline__ int visit(src_state d_src, int n, int e) const {
    const int cand = __float_as_int(d_src + edge_weight(e));
    const int r1 = claim(n, cand);
    if (!improved(r1, cand)) return 0;
    return code(r1, claim2(n), n, cand);
  }
};

__global__ void sdist_init_kernel(pipe_args a, sdist_args x, float* dist, int src_if_owned) {
  const int tid = threadIdx.x;
  const int src = src_if_owned;
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
    c->n_items[0] = src >= 0 ? 1 : 0;
    c->n_items[1] = 0;
    c->q_edges[0] = deg;
    c->q_edges[1] = 0;
    c->total_chunks = 0;
    c->edges_visited = 0;
    c->vertices_visited = 0;
    c->frontier_bitmap = 0;
    c->convert = 0;
    c->bu_R = 0;  // (plan_body would read a bottom-up level's tile ranges otherwise)
    c->bu_T = 0;
    if (src >= 0) dist[src] = 0.0f;
    x.stats_local[0] = src >= 0 ? 1 : 0;
    x.stats_local[1] = deg;
    x.stats_local[2] = 0;
    x.stats_local[3] = 0;
    a.mailbox[0] = 0;
    a.mailbox[1] = 0;
  }
}

// Head of an iteration group: global termination from the all-reduced frontier size (identical on every
// rank), local bookkeeping, chunk map.  <<<1, 1024>>>
__global__ __launch_bounds__(PLAN_BLOCK) void sdist_head_kernel(pipe_args a, sdist_args x) {
  __shared__ int s_wave[PLAN_BLOCK / 64 + 1];
  __shared__ unsigned long long s_esum[2];
  ctrl_t* c = a.ctrl;
  const int tid = threadIdx.x;
  if (c->done) return;
  if (tid == 0) {
    s_esum[0] = s_esum[1] = 0ull;
    const int level = c->level + 1;
    const int p = level & 1;
    if (x.stats_global[0] == 0) {
      c->done = 1;
      c->level = level;
      a.mailbox[1] = level;
      a.mailbox[0] = 1;
    } else {
      c->level = level;
      c->edges_visited += c->q_edges[p];      // this rank's share (sdist_stats_kernel / init)
      c->vertices_visited += c->n_items[p];
      c->n_tiles[p ^ 1] = 0;
      a.mailbox[1] = level;
    }
  }
  __syncthreads();
  if (c->done) return;
  plan_body<PLAN_BLOCK>(a, c, 1, s_wave, s_esum);
}

// Before the advance: no pending minimum for anybody.
__global__ void sdist_prep_kernel(pipe_args a, sdist_args x) {
  if (a.ctrl->done) return;
  const size_t total = (size_t)x.n_ranks * x.S;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    x.send[i] = FLT_MAX;
}

__global__ __launch_bounds__(ADV_BLOCK) void sdist_advance_kernel(pipe_args a, sssp_policy_dist pol) {
  __shared__ advance_smem<sssp_policy_dist> sm;
  ctrl_t* c = a.ctrl;
  if (c->done) return;
  pol.begin(c);
  advance_block<sssp_policy_dist, false>(a, c, pol, sm, c->level & 1, blockIdx.x, gridDim.x, c->total_chunks, a.chunk_tile);
}

// After the exchange: the smallest distance the peers found for each owned vertex; where it improves the label
// the vertex joins the next frontier -- once (the stamp tells whether this rank's own advance put it there already).
__global__ __launch_bounds__(ADV_BLOCK) void sdist_post_kernel(pipe_args a, sdist_args x, float* dist, int32_t* stamp) {
  __shared__ words_smem sm;
  ctrl_t* c = a.ctrl;
  if (c->done) return;
  const int level = c->level;
  const int q = (level + 1) & 1;
  words_to_tiles(a, c, q, x.S / 32, x.lo, [&](int w) {
    unsigned acc = 0u;
    for (int b = 0; b < 32; ++b) {
      const int off = w * 32 + b;
      const int v = x.lo + off;  // each owned vertex is looked at by exactly one thread
      if (v >= x.hi) break;
      float m = FLT_MAX;
      for (int j = 0; j < x.n_ranks; ++j)
        if (j != x.my_rank) m = fminf(m, x.recv[(size_t)j * x.S + off]);
      if (m < dist[v]) {
        dist[v] = m;
        if (stamp[v] != level) {
          stamp[v] = level;
          acc |= 1u << b;
        }
      }
    }
    return acc;
  }, sm);
}

// Size of the frontier the next iteration expands (this rank's share): the input of the all-reduce that
// drives termination.  <<<1, 1024>>>
__global__ __launch_bounds__(PLAN_BLOCK) void sdist_stats_kernel(pipe_args a, sdist_args x) {
  __shared__ unsigned long long s_red[2];
  ctrl_t* c = a.ctrl;
  const int tid = threadIdx.x;
  if (tid < 2) s_red[tid] = 0ull;
  __syncthreads();
  if (c->done) {
    if (tid < 4) x.stats_local[tid] = 0;
    return;
  }
  const int q = (c->level + 1) & 1;
  const int nt = c->n_tiles[q];
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
    atomicAdd(&s_red[0], (unsigned long long)n);
    atomicAdd(&s_red[1], (unsigned long long)m);
  }
  __syncthreads();
  if (tid == 0) {
    c->n_items[q] = (int)s_red[0];
    c->q_edges[q] = (long long)s_red[1];
    x.stats_local[0] = (long long)s_red[0];
    x.stats_local[1] = (long long)s_red[1];
    x.stats_local[2] = 0;
    x.stats_local[3] = 0;
  }
}

}  // namespace grx

using namespace grx;

struct grx_sssp_dist {
  grx_context_t ctx = nullptr;
  grx_graph_t g = nullptr;
  pipe_args a{};
  sdist_args x{};
  float* dist = nullptr;      // base pointer of the sharded labels
  int32_t* stamp = nullptr;   // base pointer of the stamps
  int grid_advance = 0, grid_post = 0;
  bool active = false;
};

extern "C" {

grx_status_t grx_sssp_dist_create(grx_context_t ctx, grx_graph_t out_rows, int32_t n_ranks, int32_t my_rank,
                                  float* d_send, const float* d_recv, long long* d_stats_local,
                                  const long long* d_stats_global, grx_sssp_dist** out) {
  if (!ctx || !out_rows || !d_send || !d_recv || !d_stats_local || !d_stats_global || !out)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp_dist_create: null argument");
  if (n_ranks < 1 || my_rank < 0 || my_rank >= n_ranks)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp_dist_create: bad rank layout");
  GRX_HIP(hipSetDevice(ctx->device));
  auto* h = new grx_sssp_dist();
  h->ctx = ctx;
  h->g = out_rows;
  const int32_t S = grx_bfs_dist_slice_bits(out_rows->V, n_ranks);
  const long long lo = (long long)my_rank * S, hi = lo + S;
  sdist_args& x = h->x;
  x.n_ranks = n_ranks;
  x.my_rank = my_rank;
  x.lo = (int32_t)(lo < out_rows->V ? lo : out_rows->V);
  x.hi = (int32_t)(hi < out_rows->V ? hi : out_rows->V);
  x.S = S;
  x.send = d_send;
  x.recv = d_recv;
  x.stats_local = d_stats_local;
  x.stats_global = d_stats_global;
  grx_status_t rc = pipeline_prepare(ctx, out_rows, &h->a);
  if (rc != GRX_SUCCESS) { delete h; return rc; }
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sdist_advance_kernel, ADV_BLOCK, 0) != hipSuccess || n < 1) n = 4;
  h->grid_advance = ctx->num_cus * (n > 8 ? 8 : n);
  h->grid_post = ctx->num_cus * 2;
  GRX_HIP(ctx->labels.reserve((size_t)S * sizeof(int32_t)));  // stamps of the owned slice
  *out = h;
  return GRX_SUCCESS;
}

// d_local: S floats, the owned slice (vertex v at d_local[v - my_rank * S]); the caller all-reduces stats_local into
// stats_global once before the first grx_sssp_dist_pre
grx_status_t grx_sssp_dist_begin(grx_sssp_dist* h, int32_t source, float* d_local) {
  if (!h || !d_local) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp_dist_begin: null argument");
  if (source < 0 || source >= h->g->V) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp_dist_begin: source out of range");
  grx_context_t ctx = h->ctx;
  GRX_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const sdist_args& x = h->x;
  h->dist = d_local - x.lo;
  h->stamp = ctx->labels.as<int32_t>() - x.lo;
  if (x.hi > x.lo) {
    GRX_HIP(fill_f32(s, d_local, FLT_MAX, x.hi - x.lo));
    GRX_HIP(fill_i32(s, ctx->labels.as<int32_t>(), -1, x.hi - x.lo));
  }
  ctx->h_mailbox[0] = 0;
  GRX_HIP(hipEventRecord(ctx->ev_begin, s));
  const int src_if_owned = (source >= x.lo && source < x.hi) ? source : -1;
  hipLaunchKernelGGL(sdist_init_kernel, dim3(1), dim3(TILE), 0, s, h->a, x, h->dist, src_if_owned);
  GRX_HIP(hipGetLastError());
  h->active = true;
  return GRX_SUCCESS;
}

// head (termination, chunk map) -> clear the outgoing minima -> advance; then the caller exchanges send -> recv
// (all_to_all_single, n_ranks equal splits of S floats)
grx_status_t grx_sssp_dist_pre(grx_sssp_dist* h) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp_dist_pre: no search in flight");
  grx_context_t ctx = h->ctx;
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(sdist_head_kernel, dim3(1), dim3(PLAN_BLOCK), 0, s, h->a, h->x);
  hipLaunchKernelGGL(sdist_prep_kernel, dim3(ctx->num_cus * 4), dim3(256), 0, s, h->a, h->x);
  sssp_policy_dist pol{h->dist, h->stamp, h->g->w, h->x.send, h->x.lo, h->x.hi, 0};
  hipLaunchKernelGGL(sdist_advance_kernel, dim3(h->grid_advance), dim3(ADV_BLOCK), 0, s, h->a, pol);
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

// apply what the peers found, then the size of the next frontier; the caller all-reduces stats_local into stats_global
grx_status_t grx_sssp_dist_post(grx_sssp_dist* h) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp_dist_post: no search in flight");
  hipStream_t s = h->ctx->stream;
  hipLaunchKernelGGL(sdist_post_kernel, dim3(h->grid_post), dim3(ADV_BLOCK), 0, s, h->a, h->x, h->dist, h->stamp);
  hipLaunchKernelGGL(sdist_stats_kernel, dim3(1), dim3(PLAN_BLOCK), 0, s, h->a, h->x);
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

grx_status_t grx_sssp_dist_poll(grx_sssp_dist* h, int32_t* done, int32_t* level) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp_dist_poll: no search in flight");
  grx_context_t ctx = h->ctx;
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, ctx->stream));
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  if (done) *done = ctx->h_ctrl->done;
  if (level) *level = ctx->h_ctrl->level;
  return GRX_SUCCESS;
}

grx_status_t grx_sssp_dist_end(grx_sssp_dist* h, grx_run_stats_t* stats) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp_dist_end: no search in flight");
  grx_context_t ctx = h->ctx;
  hipStream_t s = ctx->stream;
  GRX_HIP(hipEventRecord(ctx->ev_end, s));
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, s));
  GRX_HIP(hipEventSynchronize(ctx->ev_end));
  GRX_HIP(hipStreamSynchronize(s));
  float ms = 0;
  GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  ctx->stats = grx_run_stats_t{};
  ctx->stats.edges_visited = ctx->h_ctrl->edges_visited;      // relaxations issued by this rank
  ctx->stats.vertices_visited = ctx->h_ctrl->vertices_visited;
  ctx->stats.search_depth = ctx->h_ctrl->level;
  ctx->stats.elapsed_ms = ms;
  if (stats) *stats = ctx->stats;
  h->active = false;
  return GRX_SUCCESS;
}

grx_status_t grx_sssp_dist_destroy(grx_sssp_dist* h) {
  if (!h) return GRX_SUCCESS;
  (void)hipSetDevice(h->ctx->device);
  (void)hipStreamSynchronize(h->ctx->stream);
  delete h;
  return GRX_SUCCESS;
}

}  // extern "C"
