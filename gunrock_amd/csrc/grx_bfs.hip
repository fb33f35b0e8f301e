// grx_bfs.hip -- breadth-first search on the device-driven frontier pipeline.
//
// Reference behaviour reproduced (include/gunrock/algorithms/bfs.hxx):
//   reset   :59-69   distances = INT_MAX, distances[source] = 0
//   advance :105-128 old = atomicMin(&dist[nbr], iteration + 1); keep if improved
//   filter  :130-146 drop the invalid (-1) slots
// Result parity: depths are bit-identical (first discovery level is unique).
// MI355X implementation: the claim is an atomicOr on a visited BITMAP
// (V/8 bytes, L2-resident) instead of an atomicMin on the 4V-byte label array;
// exactly one thread wins a vertex, writes its depth with a plain store and
// emits it -- so advance and filter are one kernel and no -1 ever reaches HBM.
#include "grx_engine.hpp"

#include <climits>

namespace grx {

// VARIANT 0: bitmap claim (default).  Tuning variants kept for A/B runs
// (engine_flags bits 8..): 1 = atomicMin on the label array like upstream,
// 2 = bitmap with an agent-scope (L1-bypassing) pre-check, 3 = variant 0 that
// also counts attempted atomics into ctrl->spare[0].
template <int VARIANT>
struct bfs_policy_t {
  using src_state = int;
  int32_t* dist;
  unsigned* visited;
  int next_depth;
  ctrl_t* ctrl;

  __device__ __forceinline__ void begin(ctrl_t* c) { next_depth = c->level + 1; ctrl = c; }
  __device__ __forceinline__ src_state load_source(int) const { return 0; }
  __device__ __forceinline__ bool precheck(src_state, int n, int) const {
    if constexpr (VARIANT == 1) return dist[n] > next_depth;
    if constexpr (VARIANT == 2)
      return (__hip_atomic_load(&visited[n >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << (n & 31))) == 0u;
    return (visited[n >> 5] & (1u << (n & 31))) == 0u;
  }
  __device__ __forceinline__ bool visit(int, src_state, int n, int) const {
    if constexpr (VARIANT == 1) return next_depth < atomicMin(&dist[n], next_depth);
    if constexpr (VARIANT == 3) atomicAdd(&ctrl->spare[0], 1);
    const unsigned bit = 1u << (n & 31);
    const unsigned old = atomicOr(&visited[n >> 5], bit);
    if (old & bit) return false;
    dist[n] = next_depth;
    return true;
  }
};
using bfs_policy = bfs_policy_t<0>;

__global__ void bfs_init_kernel(pipe_args a, int32_t* dist, unsigned* visited, int src) {
  const int tid = threadIdx.x;
  int32_t* f0 = a.frontier[0];
  f0[tid] = (tid == 0) ? src : -1;
  if (tid == 0) {
    ctrl_t* c = a.ctrl;
    const int deg = a.ro[src + 1] - a.ro[src];
    a.tile_sums[0] = deg;
    a.tile_chunks[0] = (deg + CHUNK - 1) / CHUNK;
    c->level = -1;
    c->done = 0;
    c->n_tiles[0] = 1;
    c->n_items[0] = 1;
    c->n_tiles[1] = 0;
    c->n_items[1] = 0;
    c->total_chunks = 0;
    c->edges_visited = 0;
    c->vertices_visited = 0;
    c->spare[0] = 0;
    dist[src] = 0;
    visited[src >> 5] = 1u << (src & 31);
    a.mailbox[0] = 0;
    a.mailbox[1] = 0;
    a.mailbox[2] = 0;
  }
}

}  // namespace grx

using namespace grx;

extern "C" grx_status_t grx_bfs(grx_context_t ctx, grx_graph_t g, int32_t src,
                                const grx_options_t* options, int32_t* d_dist,
                                int32_t* d_pred, float* elapsed_ms) {
  (void)d_pred;  // accepted and never written, like the reference (bfs.hxx:29)
  if (!ctx || !g || !d_dist) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs: null argument");
  if (src < 0 || src >= g->V) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs: source out of range");
  grx_options_t opt;
  if (options) opt = *options; else grx_options_default(&opt);
  if (opt.advance_load_balance == GRX_LB_WORK_STEALING)
    return fail(GRX_ERROR_UNSUPPORTED, "Load balance type not supported.");

  GRX_HIP(hipSetDevice(ctx->device));
  pipe_args a;
  grx_status_t st = pipeline_prepare(ctx, g, &a);
  if (st != GRX_SUCCESS) return st;
  const size_t bm_words = ((size_t)g->V + 31) / 32;
  GRX_HIP(ctx->bitmap[0].reserve(bm_words * sizeof(unsigned)));
  unsigned* visited = ctx->bitmap[0].as<unsigned>();
  hipStream_t s = ctx->stream;

  // problem.reset() -- outside the timed region, as in the reference
  GRX_HIP(fill_i32(s, d_dist, INT_MAX, g->V));
  GRX_HIP(hipMemsetAsync(visited, 0, bm_words * sizeof(unsigned), s));

  GRX_HIP(hipEventRecord(ctx->ev_begin, s));
  hipLaunchKernelGGL(bfs_init_kernel, dim3(1), dim3(TILE), 0, s, a, d_dist, visited, src);

  bfs_policy pol{d_dist, visited, 0, nullptr};
  const int variant = (opt.engine_flags >> 8) & 3;
  const int grid = advance_grid(ctx);
  const bool profile = (opt.engine_flags & GRX_FLAG_PROFILE) != 0;
  ctx->levels.clear();
  hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
  if (profile) for (auto& e : pe) GRX_HIP(hipEventCreate(&e));

  hipError_t launch_err = hipSuccess;
  int64_t prof_v = 0, prof_e = 0;
  st = run_levels(ctx, opt, [&](hipStream_t stream, int) {
    if (profile) (void)hipEventRecord(pe[0], stream);
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, a);
    if (profile) (void)hipEventRecord(pe[1], stream);
    switch (variant) {
      case 1: hipLaunchKernelGGL((advance_kernel<bfs_policy_t<1>>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a,
                                 bfs_policy_t<1>{d_dist, visited, 0, nullptr}); break;
      case 2: hipLaunchKernelGGL((advance_kernel<bfs_policy_t<2>>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a,
                                 bfs_policy_t<2>{d_dist, visited, 0, nullptr}); break;
      case 3: hipLaunchKernelGGL((advance_kernel<bfs_policy_t<3>>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a,
                                 bfs_policy_t<3>{d_dist, visited, 0, nullptr}); break;
      default: hipLaunchKernelGGL((advance_kernel<bfs_policy>), dim3(grid), dim3(ADV_BLOCK), 0, stream, a, pol);
    }
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
  }, [&](const ctrl_t& h) {
    if (profile && !ctx->levels.empty() && !h.done) {
      level_rec& r = ctx->levels.back();
      r.frontier_size = h.vertices_visited - prof_v;
      r.edges = h.edges_visited - prof_e;
      prof_v = h.vertices_visited;
      prof_e = h.edges_visited;
    } else if (profile && h.done && !ctx->levels.empty()) {
      ctx->levels.pop_back();  // the level that only detected the empty frontier
    }
  });
  if (st != GRX_SUCCESS) return st;
  if (launch_err != hipSuccess) return fail(GRX_ERROR_HIP, hipGetErrorString(launch_err));

  GRX_HIP(hipEventRecord(ctx->ev_end, s));
  GRX_HIP(hipEventSynchronize(ctx->ev_end));
  float ms = 0;
  GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  if (profile) for (auto& e : pe) (void)hipEventDestroy(e);

  ctx->stats.edges_visited = ctx->h_ctrl->edges_visited;
  ctx->stats.vertices_visited = ctx->h_ctrl->vertices_visited;
  ctx->stats.search_depth = ctx->h_ctrl->level;
  ctx->stats.elapsed_ms = ms;
  ctx->stats.n_levels_recorded = (int32_t)ctx->levels.size();
  ctx->stats.reserved = (float)ctx->h_ctrl->spare[0];  // tuning variant 3: attempted atomics
  if (elapsed_ms) *elapsed_ms = ms;
  return GRX_SUCCESS;
}
