// grx_sssp.hip -- single-source shortest paths on the device-driven pipeline.
//
// Reference behaviour reproduced (include/gunrock/algorithms/sssp.hxx):
//   init/reset :53-80    distances = FLT_MAX, distances[source] = 0, visited = -1
//   advance    :116-130  nd = dist[src] + w; old = atomicMin(&dist[nbr], nd); keep if nd < old
//   filter     :132-151  bypass filter: drop a vertex already stamped this iteration
// It is a label-correcting (frontier Bellman-Ford) search; fl(a + w) is monotone
// in a for w >= 0, so the fixed point equals the reference CPU Dijkstra result
// bit for bit (the reference's --validate relies on the same fact).
// MI355X implementation: float atomicMin is ONE native integer atomic (ordered
// bit patterns) instead of the reference's CAS loop (cuda/atomic_functions.hxx:34-44);
// the per-iteration stamp is taken with atomicExch, so the output frontier holds
// each improved vertex exactly once per level (the reference's stamp is racy);
// advance + filter are one kernel.
#include "grx_engine.hpp"

#include <cfloat>

namespace grx {

struct sssp_policy {
  using src_state = float;
  float* dist;
  int32_t* stamp;
  const float* w;
  int level;

  __device__ __forceinline__ void begin(ctrl_t* c) { level = c->level; }
  __device__ __forceinline__ src_state load_source(int v) const { return dist[v]; }
  __device__ __forceinline__ float edge_weight(int e) const { return w ? w[e] : 1.0f; }
  __device__ __forceinline__ bool precheck(src_state d_src, int n, int e) const {
    // dist[] only decreases, so a possibly stale read can only be too large:
    // "cannot improve the stale value" implies "cannot improve the current one".
    return d_src + edge_weight(e) < dist[n];
  }
  __device__ __forceinline__ bool visit(int, src_state d_src, int n, int e) const {
    const float nd = d_src + edge_weight(e);
    const float old = dev::atomic_min_f32(&dist[n], nd);
    if (!(nd < old)) return false;
    return atomicExch(&stamp[n], level) != level;
  }
};

__global__ void sssp_init_kernel(pipe_args a, float* dist, int src) {
  const int tid = threadIdx.x;
  a.frontier[0][tid] = (tid == 0) ? src : -1;
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
    c->edges_visited = 0;
    c->vertices_visited = 0;
    c->mode = 0;
    c->q_edges[0] = 0;
    c->q_edges[1] = 0;
    dist[src] = 0.0f;
    a.mailbox[0] = 0;
  }
}

}  // namespace grx

using namespace grx;

extern "C" grx_status_t grx_sssp(grx_context_t ctx, grx_graph_t g, int32_t src,
                                 const grx_options_t* options, float* d_dist, int32_t* d_pred,
                                 float* elapsed_ms) {
  (void)d_pred;  // never written by the reference either (no store in sssp.hxx)
  if (!ctx || !g || !d_dist) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp: null argument");
  if (src < 0 || src >= g->V) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_sssp: source out of range");
  grx_options_t opt;
  if (options) opt = *options; else grx_options_default(&opt);
  if (opt.advance_load_balance == GRX_LB_WORK_STEALING)
    return fail(GRX_ERROR_UNSUPPORTED, "Load balance type not supported.");

  GRX_HIP(hipSetDevice(ctx->device));
  pipe_args a;
  grx_status_t st = pipeline_prepare(ctx, g, &a);
  if (st != GRX_SUCCESS) return st;
  GRX_HIP(ctx->labels.reserve((size_t)g->V * sizeof(int32_t)));
  int32_t* stamp = ctx->labels.as<int32_t>();
  hipStream_t s = ctx->stream;

  // problem.init()/reset(), outside the timed region as in the reference
  GRX_HIP(fill_f32(s, d_dist, FLT_MAX, g->V));
  GRX_HIP(fill_i32(s, stamp, -1, g->V));

  GRX_HIP(hipEventRecord(ctx->ev_begin, s));
  hipLaunchKernelGGL(sssp_init_kernel, dim3(1), dim3(TILE), 0, s, a, d_dist, src);

  sssp_policy pol{d_dist, stamp, g->w, 0};
  const int grid = advance_grid(ctx);
  ctx->levels.clear();
  hipError_t launch_err = hipSuccess;
  st = run_levels(ctx, opt, [&](hipStream_t stream, int) {
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, a, 0);
    hipLaunchKernelGGL((advance_kernel<sssp_policy>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a, pol);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) launch_err = e;
  }, [&](const ctrl_t&) {});
  if (st != GRX_SUCCESS) return st;
  if (launch_err != hipSuccess) return fail(GRX_ERROR_HIP, hipGetErrorString(launch_err));

  GRX_HIP(hipEventRecord(ctx->ev_end, s));
  GRX_HIP(hipEventSynchronize(ctx->ev_end));
  float ms = 0;
  GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  ctx->stats.edges_visited = ctx->h_ctrl->edges_visited;
  ctx->stats.vertices_visited = ctx->h_ctrl->vertices_visited;
  ctx->stats.search_depth = ctx->h_ctrl->level;
  ctx->stats.elapsed_ms = ms;
  ctx->stats.n_levels_recorded = 0;
  if (elapsed_ms) *elapsed_ms = ms;
  return GRX_SUCCESS;
}
