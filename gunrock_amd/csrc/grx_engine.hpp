// grx_engine.hpp -- host-side driver of the device-driven level loop
// (the MI355X counterpart of enactor_t::enact(), framework/enactor.hxx:243-288).
#pragma once

#include "grx_frontier.hpp"

#include <chrono>
#include <cstdlib>

namespace grx {

// Fill kernels (problem_t::reset(): algorithms/bfs.hxx:59-69, sssp.hxx:63-80).
__global__ void fill_i32_kernel(int32_t* p, int32_t value, int64_t n);
__global__ void fill_f32_kernel(float* p, float value, int64_t n);

hipError_t fill_i32(hipStream_t s, int32_t* p, int32_t value, int64_t n);
hipError_t fill_f32(hipStream_t s, float* p, float value, int64_t n);

// Size all pipeline scratch for graph g; fills `a`.
grx_status_t pipeline_prepare(grx_context_t ctx, grx_graph_t g, pipe_args* a);

// Build (once) and cache the transpose of g in the graph handle.
grx_status_t graph_build_transpose(grx_context_t ctx, grx_graph_t g);

// Partitioned searches: a copy of g's column array with every row's hub entries first (same offsets), built once per handle
// into g->hf_ci (hf_state 1) -- or not at all (hf_state 2: no room, switched off) -- see grx_transpose.hip.
grx_status_t graph_build_hub_first(grx_context_t ctx, grx_graph_t g);

// Verify (once per graph handle, cached) that the CSR equals its transpose: the caller-supplied
// `symmetric` property is only trusted after this check (direction-optimising BFS uses the CSR
// itself as the in-edge list of a symmetric graph).
grx_status_t graph_is_symmetric(grx_context_t ctx, grx_graph_t g, bool* result);

// Edge-weight statistics, once per graph handle (weight_sum / weight_min / weight_max / uniform_weights).
grx_status_t graph_weight_stats(grx_context_t ctx, grx_graph_t g);
// true when the graph needs no weight stream: no values array, or every weight exactly 1.0
// (graph_weight_stats must have run)
inline bool graph_unit_weights(grx_graph_t g) {
  return !g->w || (g->weight_sum >= 0.0 && g->uniform_weights && g->weight_min == 1.0f);
}

// Block-asynchronous relaxation for road-like graphs (grx_block.hip): is it to be used for this graph (weighted: float
// labels relaxed with the edge weights; else BFS depths)?  Builds the per-graph block structure on first use.
grx_status_t blk_prepare(grx_context_t ctx, grx_graph_t g, bool weighted, bool* usable);
grx_status_t blk_search(grx_context_t ctx, grx_graph_t g, int32_t src, const grx_options_t& opt, bool weighted, void* d_out,
                        float* elapsed_ms);
void blk_graph_free(void* p);

// GRX_PREP_TIMING=1: wall time of every per-graph preprocessing step (stream drained at both ends) on stderr --
// what `first_call_ms` of bench.py is made of.
struct prep_timer {
  const char* what;
  hipStream_t s;
  bool on;
  std::chrono::steady_clock::time_point t0;
  prep_timer(const char* w, hipStream_t stream) : what(w), s(stream) {
    const char* v = getenv("GRX_PREP_TIMING");
    on = v && *v == '1';
    if (on) {
      (void)hipStreamSynchronize(s);
      t0 = std::chrono::steady_clock::now();
    }
  }
  ~prep_timer() {
    if (!on) return;
    (void)hipStreamSynchronize(s);
    fprintf(stderr, "[grx prep] %-40s %9.3f ms\n", what,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

// Launch configuration of the advance kernel (persistent workgroups striding over chunks).
// Upper bound used for sizing scratch:
inline int advance_grid(grx_context_t ctx) { return ctx->num_cus * 8; }
// Low-degree (road-like) graphs have thousands of tiny levels: a level is a handful of
// chunks, and dispatching 2048 idle workgroups per launch would dominate.  One workgroup
// per CU is plenty there; scale-free graphs get the full 8 per CU.
inline int advance_grid_for(grx_context_t ctx, grx_graph_t g) {
  const bool road_like = g->V > 0 && (long long)g->E < 4ll * g->V;
  return road_like ? ctx->num_cus * (GRX_MID_WGS / 32) : ctx->num_cus * 8;
}

// Generic host loop: `launch_level(stream, i)` enqueues level group i (head + level kernels);
// the host never passes a level-dependent size.  Two pacing modes:
//  * batches (pace_depth == 0): groups are enqueued blindly in growing batches and the host
//    reads the control block between batches -- right for thousands of short levels;
//  * paced (pace_depth > 0): the head kernel of group i publishes i in the pinned mailbox
//    (word 3) and `done` in word 0; the host keeps at most `pace_depth` groups queued behind
//    the running one and stops enqueueing the moment `done` shows up.  No blocking round trip
//    per level, and at most pace_depth wasted (no-op) groups at the end -- right for the few
//    fat levels of a scale-free search, where a wasted full-grid group costs ~9 us.
//    The caller resets mailbox words 0 and 3 (host side) before the first launch.
//    fast_return (paced mode only): the kernel that finds the search finished publishes the final
//    counters in the mailbox before the done flag (publish_done), so the host returns the moment
//    it sees the flag -- no device-to-host copy, no stream synchronisation; the (at most
//    pace_depth) groups still queued exit on `done` behind the caller's back, and anything the
//    caller enqueues next on this stream is ordered after them.  *returned_fast tells the caller
//    that ctx->h_ctrl was filled from the mailbox.
//    hold_after (paced mode only, 0: off): the number of groups the PREVIOUS search on the graph needed.  The no-op
//    groups queued behind the end of a search are what a back-to-back sequence of searches pays for pacing (two
//    groups = four full-grid launches, ~18 us of a 160 us search): with a prediction the host enqueues exactly
//    hold_after groups and then waits -- for `done`, or for the stream to drain without it (the search is longer than
//    the last one: a bubble of one host round trip, once, then the usual pacing).  *groups_used receives the number
//    of groups that started before the end was published (the next search's hold_after).
//    batch_after_first (batch mode, 0: off): first_batch is a prediction (the groups the previous search needed); when it
//    falls short the batches restart from this size instead of doubling the prediction.
template <class LaunchLevel, class AfterSync>
grx_status_t run_levels(grx_context_t ctx, const grx_options_t& opt, LaunchLevel launch_level,
                        AfterSync after_sync, int first_batch = 4, int pace_depth = 0, bool fast_return = false,
                        bool* returned_fast = nullptr, int hold_after = 0, int* groups_used = nullptr,
                        int batch_after_first = 0) {
  if (returned_fast) *returned_fast = false;
  if (groups_used) *groups_used = 0;
  const bool sync_each = (opt.engine_flags & (GRX_FLAG_SYNC_EACH_LEVEL | GRX_FLAG_PROFILE)) != 0;
  int batch = sync_each ? 1 : first_batch;
  int launched = 0;
  hipGraphExec_t graph_exec = nullptr;
  bool graph_failed = false;
  static const bool use_graph = [] { const char* v = getenv("GRX_USE_GRAPH"); return !(v && *v == '0'); }();
  struct graph_guard {
    hipGraphExec_t& e;
    ~graph_guard() { if (e) (void)hipGraphExecDestroy(e); }
  } guard{graph_exec};
  const int max_levels = opt.max_iterations > 0 ? opt.max_iterations : 0x7fffffff;
  if (sync_each) pace_depth = 0;
  for (;;) {
    if (pace_depth > 0) {
      volatile int32_t* mb = ctx->h_mailbox;
      unsigned spins = 0;
      for (;;) {
        if (mb[0] != 0) {  // done
          if (groups_used) *groups_used = mb[3] + 1;
          if (fast_return) {
            const volatile long long* mb64 = reinterpret_cast<const volatile long long*>(mb + 4);
            ctx->h_ctrl->done = 1;
            ctx->h_ctrl->level = mb[1];
            ctx->h_ctrl->edges_visited = mb64[0];
            ctx->h_ctrl->vertices_visited = mb64[1];
            ctx->mailbox_ticks = mb64[2];
            if (returned_fast) *returned_fast = true;
            after_sync(*ctx->h_ctrl);
            return GRX_SUCCESS;
          }
          break;
        }
        const int started = mb[3] + 1;
        const bool hold = hold_after > 0 && launched == hold_after;  // every predicted group is queued
        if (launched < max_levels && launched - started < pace_depth && !hold) {
          launch_level(ctx->stream, launched);
          ++launched;
          spins = 0;
          continue;
        }
        if (launched >= max_levels && started >= launched) break;
        __builtin_ia32_pause();
        ++spins;
        if (hold && started >= launched && (spins & 0x3f) == 0) {
          // the last predicted group has started and the end has not been published yet: still running, or is this
          // search longer than the previous one?  (A drained stream has made its mailbox writes visible.)
          const hipError_t q = hipStreamQuery(ctx->stream);
          if (q == hipSuccess) {
            if (mb[0] == 0) { hold_after = 0; spins = 0; }
            continue;
          }
          if (q != hipErrorNotReady) GRX_HIP(q);
          (void)hipGetLastError();
        }
        if ((spins & 0xfffff) == 0) {
          // no progress for a long time: if the stream drained without `done` (mailbox writes
          // not visible to this host?), fall back to blind enqueueing
          hipError_t q = hipStreamQuery(ctx->stream);
          if (q == hipSuccess) {
            if (mb[0] != 0) break;
            for (int i = 0; i < pace_depth && launched < max_levels; ++i, ++launched) launch_level(ctx->stream, launched);
            break;
          }
          if (q != hipErrorNotReady) GRX_HIP(q);
        }
      }
    } else {
      // High-diameter searches (thousands of short levels) are bound by the HOST's launch rate:
      // two launches per level at ~3.5-5 us each against ~10 us of device time.  No kernel takes a
      // level-dependent argument, so once a search has outlived its first 64 groups the next
      // GRAPH_GROUPS groups are captured into a hipGraph ONCE and every later batch is a few graph
      // replays (one host call per 64 levels).  Capture + instantiate costs about what launching
      // the same groups costs, so short searches never pay for it.
      constexpr int GRAPH_GROUPS = 64;
      if (!sync_each && use_graph && !graph_exec && !graph_failed && launched >= 64 && batch >= GRAPH_GROUPS &&
          launched + GRAPH_GROUPS <= max_levels) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          for (int i = 0; i < GRAPH_GROUPS; ++i) launch_level(ctx->stream, launched + i);
          if (hipStreamEndCapture(ctx->stream, &graph) != hipSuccess || !graph ||
              hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            graph_exec = nullptr;
            graph_failed = true;
          }
          if (graph) (void)hipGraphDestroy(graph);
        } else {
          graph_failed = true;
        }
        (void)hipGetLastError();
      }
      if (graph_exec) {
        int left = batch;
        while (left >= GRAPH_GROUPS && launched + GRAPH_GROUPS <= max_levels) {
          GRX_HIP(hipGraphLaunch(graph_exec, ctx->stream));
          launched += GRAPH_GROUPS;
          left -= GRAPH_GROUPS;
        }
        // (what is left of a batch that is not a multiple of the graph -- a batch restarted after a predicted first batch
        // fell short, or the last groups before max_levels -- is launched directly)
        for (; left > 0 && launched < max_levels; --left, ++launched) launch_level(ctx->stream, launched);
      } else {
        for (int i = 0; i < batch && launched < max_levels; ++i, ++launched) launch_level(ctx->stream, launched);
      }
    }
    GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, ctx->stream));
    GRX_HIP(hipStreamSynchronize(ctx->stream));
    after_sync(*ctx->h_ctrl);
    if (ctx->h_ctrl->done) break;
    if (launched >= max_levels) break;
    if (!sync_each && batch_after_first > 0) {
      batch = batch_after_first;
      batch_after_first = 0;
    } else if (!sync_each && batch < (graph_exec ? 1024 : 64)) {
      batch *= 2;
    }
  }
  return GRX_SUCCESS;
}

}  // namespace grx
