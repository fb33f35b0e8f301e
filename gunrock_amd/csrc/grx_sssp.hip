// grx_sssp.hip -- single-source shortest paths on the device-driven pipeline.
//
// Reference behaviour reproduced (include/gunrock/algorithms/sssp.hxx):
//   init/reset :53-80    distances = FLT_MAX, distances[source] = 0, visited = -1
//   advance    :116-130  nd = dist[src] + w; old = atomicMin(&dist[nbr], nd); keep if nd < old
//   filter     :132-151  bypass filter: drop a vertex already stamped this iteration
// It is a label-correcting (frontier Bellman-Ford) search; fl(a + w) is monotone in a for
// w >= 0, so the fixed point equals the reference CPU Dijkstra result bit for bit (the
// reference's --validate relies on the same fact) -- whatever the relaxation schedule.
//
// MI355X implementation
//  * float atomicMin is ONE native integer atomic on the ordered bit pattern instead of the
//    reference's CAS loop (cuda/atomic_functions.hxx:34-44); a stale pre-check skips
//    hopeless atomics; the per-iteration stamp is taken with atomicExch, so the output
//    frontier holds each improved vertex exactly once; advance + filter are one kernel.
//  * near-far schedule (delta-stepping, Davidson et al.; the reference lists `bucketing`
//    as an empty stub, operators/advance/bucketing.hxx:30-35): improved vertices whose
//    tentative distance falls in the current bucket [lo, hi) go to the next frontier, the
//    others to a FAR pile (side output of the advance kernel, flushed through LDS with one
//    reservation atomic per flush).  When the frontier drains, the head kernel moves the
//    bucket on -- jumping over empty buckets with the tracked minimum of the pile -- and
//    the level kernel pulls the new bucket out of the pile (sssp_split_body).  On weighted road-like graphs
//    this cuts the relaxations of plain label-correcting by one to two orders of magnitude;
//    the distances are identical.  Unit-weight graphs and engine_flags bit 4 use the plain
//    schedule.
#include "grx_engine.hpp"
#include "grx_mid.hpp"
#include "grx_relax.hpp"

#include <cfloat>
#include <cmath>
#include <climits>
#include <cstdlib>
#include <vector>

namespace grx {

struct sssp_policy {
  using src_state = float;
  float* dist;
  int32_t* stamp;
  const float* w;
  int level;
  int l2_local;  // set by mid_levels_body: the launch sits on one XCD, atomics may execute in its L2
  // all weights equal: every tentative distance of a level is the same number, so exactly ONE relaxation per vertex
  // ever finds `nd < old` (the first; its label is final) -- the output needs no per-level stamp to be duplicate-free
  int uniform;

  __device__ __forceinline__ void begin(ctrl_t* c) { level = c->level; }
  __device__ __forceinline__ void set_level(int l) { level = l; }
  // past the CU's L1: the label may have been lowered by an atomic (performed at L2) of
  // this very launch (multi-level kernel), and relaxing from a stale label would lose it
  __device__ __forceinline__ src_state load_source(int v) const {
    return __hip_atomic_load(&dist[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // equal weights: the label a vertex is claimed with is final, so it travels with the queue entry (grx_mid.hpp)
  __device__ __forceinline__ bool carry_state() const { return uniform != 0; }
  __device__ __forceinline__ src_state state_from_bits(int b) const { return __int_as_float(b); }
  __device__ __forceinline__ float edge_weight(int e) const { return w ? w[e] : 1.0f; }
  static constexpr bool two_claims = true;
  __device__ __forceinline__ bool precheck(src_state d_src, int n, int e, int& cand) const {
    // dist[] only decreases, so a possibly stale read can only be too large:
    // "cannot improve the stale value" implies "cannot improve the current one".
    const float nd = d_src + edge_weight(e);
    cand = __float_as_int(nd);
    return nd < dist[n];
  }
  // what precheck computes, without the probe (mid_levels_body)
  __device__ __forceinline__ bool prepare(src_state d_src, int, int e, int& cand) const {
    cand = __float_as_int(d_src + edge_weight(e));
    return true;
  }
  __device__ __forceinline__ int claim(int n, int cand) const {
    // tentative distances are >= 0: the integer order of the bit patterns is the float order
    if (l2_local)
      return __hip_atomic_fetch_min(reinterpret_cast<int*>(&dist[n]), cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __float_as_int(dev::atomic_min_f32(&dist[n], __int_as_float(cand)));
  }
  __device__ __forceinline__ bool improved(int raw1, int cand) const { return __int_as_float(cand) < __int_as_float(raw1); }
  __device__ __forceinline__ bool need2(int raw1, int cand) const { return !uniform && improved(raw1, cand); }
  // once per level: the iteration stamp de-duplicates the output exactly
  __device__ __forceinline__ int claim2(int n) const {
    if (l2_local) return __hip_atomic_exchange(&stamp[n], level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return atomicExch(&stamp[n], level);
  }
  __device__ __forceinline__ int code(int raw1, int raw2, int, int cand) const {
    return (improved(raw1, cand) && (uniform || raw2 != level)) ? 1 : 0;
  }
  __device__ __forceinline__ int visit(src_state d_src, int n, int e) const {
    const int cand = __float_as_int(d_src + edge_weight(e));
    const int r1 = claim(n, cand);
    if (!improved(r1, cand)) return 0;
    return code(r1, uniform ? 0 : claim2(n), n, cand);
  }
};

struct sssp_nf_args {
  int32_t* far[2];
  int32_t capacity;
};

struct sssp_nf_policy {
  using src_state = float;
  static constexpr bool has_side = true;
  static constexpr bool drained_is_done = false;  // an empty frontier means "bucket drained" (mid_levels_body)
  float* dist;
  int32_t* stamp;
  const float* w;
  sssp_nf_args nf;
  int level;
  float hi;
  int32_t* far_out;
  unsigned* min_far;
  const int32_t* refill_in;  // the pile a bucket is being pulled from (refill_rebind)
  int l2_local;  // set by mid_levels_body: the launch sits on one XCD, label / stamp atomics may execute in its L2
  int sel;       // the far pile that receives appends (ctrl.nf_sel as of begin / refill_rebind)
  float lo;      // lower bound of the current bucket

  __device__ __forceinline__ void set_level(int l) { level = l; }

  __device__ __forceinline__ void begin(ctrl_t* c) {
    level = c->level;
    hi = c->nf_hi;
    lo = c->nf_lo;
    sel = c->nf_sel;
    // (a select, not far[sel]: a dynamically indexed member forces the whole by-value policy into memory -- the compiler
    // then keeps one copy per thread in LDS, 20 KB, which pushed this kernel past 64 KB of LDS: see sssp_nf_level_kernel)
    far_out = sel ? nf.far[1] : nf.far[0];
    min_far = &c->nf_min_far;
  }

  // ---- the NEXT BUCKET inside a many-levels launch (grx_mid.hpp, `policy_refills`; round 6) ------------------------------
  // A road-network search changes bucket ~1000 times, and every change used to end the launch: head kernel (bucket
  // bookkeeping) -> level kernel that only pulls the bucket out of the pile -> head kernel (plan) -> level kernel, ~30 us of
  // launches and heads around ~6 levels of ~9 us.  Inside the launch it is the leader's bookkeeping, one exchange to publish
  // it, the pile dealt out in blocks of 256 entries to the resident workgroups, and the exchange every level ends with.
  // Everything the launch shares lives in the home XCD's L2 (workgroup-scope atomics, loads past the L1), as the labels do.
  // One thread of the leader.  Returns false: leave it to the head kernel (pile empty = end of the search, pile too large for
  // 32 workgroups, overflow).  What it writes is what sssp_nf_head_kernel writes.
  __device__ __forceinline__ bool refill_decide(ctrl_t* c, int max_n) {
    const int far_n = __hip_atomic_load(&c->nf_far_n[sel], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (far_n <= 0 || far_n > max_n || far_n > nf.capacity) return false;
    const float delta = c->nf_delta;  // (constant during a search)
    const float closest = __uint_as_float(__hip_atomic_load(&c->nf_min_far, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const float lo2 = hi;
    float hi2 = lo2 + delta;
    if (closest >= hi2 && closest < FLT_MAX) {
      hi2 = (floorf(closest / delta) + 1.0f) * delta;
      if (!(hi2 > closest)) hi2 = nextafterf(closest, FLT_MAX);
    }
    if (!(hi2 > lo2)) hi2 = nextafterf(lo2, FLT_MAX);
    __hip_atomic_store(&c->nf_lo, lo2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&c->nf_hi, hi2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&c->nf_min_far, 0x7f7fffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&c->nf_far_n[sel ^ 1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&c->nf_sel, sel ^ 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)__hip_atomic_fetch_add(&c->nf_phases, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return true;
  }
  // every thread, behind the exchange that published the decision: the new bucket, past the L1.  Returns the entries of the
  // pile the bucket is pulled from (the one that received appends so far).
  __device__ __forceinline__ int refill_rebind(ctrl_t* c) {
    const int n = __hip_atomic_load(&c->nf_far_n[sel], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    refill_in = far_out;
    lo = __hip_atomic_load(&c->nf_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hi = __hip_atomic_load(&c->nf_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sel ^= 1;
    far_out = sel ? nf.far[1] : nf.far[0];
    return min(n, nf.capacity);
  }
  __device__ __forceinline__ int refill_entry(int i) const {
    return __hip_atomic_load(&refill_in[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // 1: the label lies in the bucket -> frontier; 2: beyond it -> stays in the (new) pile, key = its ordered bits; 0: stale
  // (sssp_split_body)
  __device__ __forceinline__ int refill_class(src_state dv, unsigned& key) const {
    key = __float_as_uint(dv);
    return dv >= hi ? 2 : (dv >= lo ? 1 : 0);
  }
  __device__ __forceinline__ void refill_keep_store(int i, int v) const { far_out[i] = v; }
  // past the L1: inside a multi-level launch the label may have been lowered by an atomic (performed in L2) since
  // this CU last read the line
  __device__ __forceinline__ src_state load_source(int v) const {
    return __hip_atomic_load(&dist[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ float edge_weight(int e) const { return w[e]; }
  static constexpr bool two_claims = true;
  __device__ __forceinline__ bool precheck(src_state d_src, int n, int e, int& cand) const {
    const float nd = d_src + edge_weight(e);
    cand = __float_as_int(nd);
    return nd < dist[n];
  }
  __device__ __forceinline__ bool prepare(src_state d_src, int, int e, int& cand) const {
    cand = __float_as_int(d_src + edge_weight(e));
    return true;
  }
  __device__ __forceinline__ int claim(int n, int cand) const {
    if (l2_local)  // tentative distances are >= 0: integer order of the bit patterns == float order
      return __hip_atomic_fetch_min(reinterpret_cast<int*>(&dist[n]), cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __float_as_int(dev::atomic_min_f32(&dist[n], __int_as_float(cand)));
  }
  // improved AND inside the bucket: joins the next frontier, once per level (stamp)
  __device__ __forceinline__ bool need2(int raw1, int cand) const {
    const float nd = __int_as_float(cand);
    return nd < __int_as_float(raw1) && nd < hi;
  }
  __device__ __forceinline__ int claim2(int n) const {
    if (l2_local) return __hip_atomic_exchange(&stamp[n], level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return atomicExch(&stamp[n], level);
  }
  __device__ __forceinline__ int code(int raw1, int raw2, int, int cand) const {
    const float nd = __int_as_float(cand), old = __int_as_float(raw1);
    if (!(nd < old)) return 0;
    if (nd < hi) return raw2 != level ? 1 : 0;
    // beyond the bucket.  A finite previous label >= hi means the vertex already waits in
    // the far pile (it was appended when that label was set, and entries only leave the
    // pile once their label drops below a bucket bound): nothing to add.  Exactly one of
    // several concurrent first relaxations sees FLT_MAX and appends.
    if (old != FLT_MAX && old >= hi) return 0;
    return 2;
  }
  __device__ __forceinline__ int visit(src_state d_src, int n, int e) const {
    const int cand = __float_as_int(d_src + edge_weight(e));
    const int r1 = claim(n, cand);
    return code(r1, need2(r1, cand) ? claim2(n) : 0, n, cand);
  }
  // one reservation atomic per flush of the workgroup's side buffer
  __device__ __forceinline__ int side_reserve(ctrl_t* c, int n) const {
    // (inside mid_levels_body only workgroups of ONE XCD touch the pile counters: the atomic may execute in its L2)
    const int base = l2_local ? __hip_atomic_fetch_add(&c->nf_far_n[sel], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                              : atomicAdd(&c->nf_far_n[sel], n);
    if (base + n > nf.capacity) {
      c->nf_overflow = 1;
      return -1;
    }
    return base;
  }
  // returns the ordered bits of the stored vertex's label; the caller reduces them over the
  // wave and commits ONE atomicMin (a single word sustains only ~90 atomics/us)
  __device__ __forceinline__ unsigned side_store(int i, int v) const {
    far_out[i] = v;
    return __float_as_uint(dist[v]);
  }
  // lower bound of the labels waiting in the pile (may go stale; only steers how far
  // the bucket jumps, never what is dropped)
  __device__ __forceinline__ void side_commit(unsigned key_min) const {
    if (l2_local) (void)__hip_atomic_fetch_min(min_far, key_min, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicMin(min_far, key_min);
  }
};

// The near-far policy INSIDE a many-levels launch (grx_mid.hpp), round 6: ONE atomic per relaxed edge and no label load per
// expanded vertex.  The level-synchronous kernels let a vertex join the next frontier once per level (second atomic: the stamp)
// and read its label when they expand it (it may have dropped again after it joined).  Here EVERY improving relaxation appends
// {vertex, the label it wrote}: a vertex improved twice in one level is expanded twice, the second time from the lower label --
// the relaxations of the stale entry are no-ops or short-lived, the fixed point is the same -- and in exchange a level's chain
// loses two dependent round trips (the stamp exchange behind the atomicMin, the label load behind the queue entry): on the
// weighted road stand-in 13.4 us per level with both.  The stamps stay as the level-synchronous kernels left them (older
// levels), so a search may move between the two kinds of kernels in both directions.
// MEASURED AND NOT ADOPTED (-DGRX_NF_CARRY builds it): +31 % relaxations (158.5 M against 121.2 M on the weighted road stand-in)
// for 1.6 % of the time (97.3 against 98.9 ms) -- profiles/r6_c10_*.
struct sssp_nf_mid_policy : sssp_nf_policy {
  static constexpr bool two_claims = false;
  __device__ __forceinline__ bool carry_state() const { return true; }
  __device__ __forceinline__ src_state state_from_bits(int b) const { return __int_as_float(b); }
  __device__ __forceinline__ int code(int raw1, int, int, int cand) const {
    const float nd = __int_as_float(cand), old = __int_as_float(raw1);
    if (!(nd < old)) return 0;
    if (nd < hi) return 1;
    if (old != FLT_MAX && old >= hi) return 0;  // (already waits in the far pile: sssp_nf_policy::code)
    return 2;
  }
};

__global__ void sssp_init_kernel(pipe_args a, float* dist, int src, float delta) {
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
    c->t_start = (long long)wall_clock64();
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
    c->nf_lo = 0.0f;
    c->nf_hi = delta;
    c->nf_delta = delta;
    c->nf_min_far = 0x7f7fffffu;  // FLT_MAX
    c->nf_far_n[0] = c->nf_far_n[1] = 0;
    c->nf_sel = 0;
    c->nf_split = 0;
    c->nf_overflow = 0;
    c->nf_phases = 0;
    c->map_level = -2;
    c->bin_want = 0;
    dist[src] = 0.0f;
    a.mailbox[0] = 0;
  }
}

// Head of an iteration of the near-far schedule, ONE launch of one workgroup  <<<1, 1024>>>:
// bucket bookkeeping, then the chunk map of the level (plan_body).
//   frontier non-empty: go on inside the current bucket;
//   frontier empty: move to the next bucket -- jumping over empty ones with the tracked lower
//     bound of the waiting labels -- or finish when the far pile is empty too.  Moving on
//     sets nf_split: THIS iteration's level kernel only pulls the new bucket out of the pile
//     (sssp_split_body), and the next head (which finds nf_split set) plans that rebuilt
//     frontier under the same level number.  A bucket change therefore costs one extra
//     two-kernel group -- there are ~100 of them in a road-network search -- and every other
//     iteration is two launches instead of four (phase, split, plan, advance).
__global__ __launch_bounds__(PLAN_BLOCK) void sssp_nf_head_kernel(pipe_args a, sssp_nf_args nf, int mid_v, int mid_e) {
  __shared__ unsigned long long s_n;
  __shared__ unsigned long long s_esum[2];
  __shared__ int s_wave[PLAN_BLOCK / 64 + 1];
  __shared__ int s_go;
  ctrl_t* c = a.ctrl;
  const int tid = threadIdx.x;
  const ctrl_head h = load_ctrl_head(c);  // one batch of loads, together with nf_split
  const int resume = c->nf_split;  // the previous group rebuilt the frontier of c->level from the pile
  const int done = h.done;
  const int level = resume ? h.level : h.level + 1;
  const int p = level & 1;
  const int nt = h.nt(p);
  if (tid == 0) { s_n = 0; s_esum[0] = s_esum[1] = 0; s_go = 0; }
  __syncthreads();
  if (done) return;
  if (!resume) {
    long long n = 0;
    for (int i = tid; i < nt; i += PLAN_BLOCK) n += a.tile_count[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (dev::lane_id() == 0 && n) atomicAdd(&s_n, (unsigned long long)n);
    __syncthreads();
  }
  if (tid == 0) {
    const long long n_f = (long long)s_n;
    const int sel = c->nf_sel;
    const int far_n = min(c->nf_far_n[sel], nf.capacity);
    c->nf_split = 0;
    if (resume) {
      s_go = 1;
    } else {
      c->level = level;
      c->n_tiles[p ^ 1] = 0;
      a.mailbox[1] = level;
      if (n_f > 0) {
        s_go = 1;
      } else if (far_n == 0) {
        c->done = 1;
        publish_done(a, c, level);
      } else {
        // New bucket [old hi, hi').  Everything in the pile whose label is below the OLD hi has
        // been in a frontier since that label was set (the relaxation that produced it took the
        // `nd < hi` branch), so it may be dropped; everything below hi' becomes the frontier.
        const float delta = c->nf_delta;
        const float closest = __uint_as_float(c->nf_min_far);
        const float lo = c->nf_hi;
        float hi = lo + delta;
        if (closest >= hi && closest < FLT_MAX) {
          hi = (floorf(closest / delta) + 1.0f) * delta;
          if (!(hi > closest)) hi = nextafterf(closest, FLT_MAX);  // fp guard for huge labels
        }
        if (!(hi > lo)) hi = nextafterf(lo, FLT_MAX);
        c->nf_lo = lo;
        c->nf_hi = hi;
        c->nf_min_far = 0x7f7fffffu;  // rebuilt by the split (kept entries) and by later appends
        c->nf_split = 1;
        c->nf_sel = sel ^ 1;
        c->nf_far_n[sel ^ 1] = 0;
        c->n_tiles[p] = 0;  // the frontier of this level is rebuilt from the pile
        c->total_chunks = 0;
        c->nf_phases += 1;
      }
    }
  }
  __syncthreads();
  if (s_go) {
    plan_in in;
    in.done = 0;
    in.level = level;
    in.nt = nt;  // resume: the tiles the split emitted last launch; otherwise the advance's output
    in.mode = 0;
    in.R = 0;
    in.T = 0;
    in.mid_v = mid_v;  // a small frontier: the level kernel drains the whole bucket in one launch (grx_mid.hpp)
    in.mid_e = mid_e;
    plan_body<PLAN_BLOCK>(a, c, 2, s_wave, s_esum, in);
  }
}

// Pull the bucket [lo, hi) out of the far pile: entries whose CURRENT label lies in the
// bucket become the frontier (once each), entries beyond it are kept, the rest are stale.
struct split_smem {
  int out[2 * TILE];
  int keep[2 * ADV_BLOCK];
  int wave[ADV_BLOCK / 64 + 1];
  int res[3];
  int cnt, kcnt, kbase;
};

__device__ __forceinline__ void sssp_split_body(const pipe_args& a, const sssp_nf_args& nf, const float* dist,
                                                split_smem& sm) {
  ctrl_t* c = a.ctrl;
  const int level = c->level;
  const int p = level & 1;
  const int in_sel = c->nf_sel ^ 1;
  const int32_t* fin = in_sel ? nf.far[1] : nf.far[0];
  int32_t* fout = in_sel ? nf.far[0] : nf.far[1];
  const int n = min(c->nf_far_n[in_sel], nf.capacity);
  const float lo = c->nf_lo, hi = c->nf_hi;
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  unsigned kept_min = 0x7f7fffffu;
  if (tid == 0) { sm.cnt = 0; sm.kcnt = 0; sm.res[0] = 0; sm.res[1] = 0; }
  __syncthreads();
  for (int base = blockIdx.x * ADV_BLOCK; base < n; base += gridDim.x * ADV_BLOCK) {
    const int i = base + tid;
    bool near = false, keep = false;
    int v = -1;
    if (i < n) {
      v = fin[i];
      const float dv = dist[v];  // a vertex has at most one entry in the pile
      if (dv >= hi) { keep = true; kept_min = min(kept_min, __float_as_uint(dv)); }
      else if (dv >= lo) near = true;
    }
    const unsigned long long mn = dev::ballot(near);
    if (mn) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&sm.cnt, __popcll(mn));
      at = __shfl(at, 0, 64);
      if (near) sm.out[at + dev::mask_rank(mn)] = v;
    }
    const unsigned long long mk = dev::ballot(keep);
    if (mk) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&sm.kcnt, __popcll(mk));
      at = __shfl(at, 0, 64);
      if (keep) sm.keep[at + dev::mask_rank(mk)] = v;
    }
    __syncthreads();
    int have = sm.cnt;
    const int kc = sm.kcnt;
    __syncthreads();
    if (have >= TILE) {
      emit_tile(a, c, p, sm.out, have - TILE, TILE, sm.wave, sm.res);
      have -= TILE;
      __syncthreads();
    }
    if (kc >= ADV_BLOCK) {
      if (tid == 0) {
        int b = atomicAdd(&c->nf_far_n[in_sel ^ 1], kc);
        if (b + kc > nf.capacity) { c->nf_overflow = 1; b = -1; }
        sm.kbase = b;
      }
      __syncthreads();
      if (sm.kbase >= 0)
        for (int k = tid; k < kc; k += ADV_BLOCK) fout[sm.kbase + k] = sm.keep[k];
      __syncthreads();
    }
    if (tid == 0) {
      sm.cnt = have;
      if (kc >= ADV_BLOCK) sm.kcnt = 0;
    }
    __syncthreads();
  }
  const int rem = sm.cnt;
  if (rem > 0) emit_tile(a, c, p, sm.out, 0, rem, sm.wave, sm.res);
  __syncthreads();
  release_tiles(a, sm.res);
  const int kc = sm.kcnt;
  if (kc > 0) {
    if (tid == 0) {
      int b = atomicAdd(&c->nf_far_n[in_sel ^ 1], kc);
      if (b + kc > nf.capacity) { c->nf_overflow = 1; b = -1; }
      sm.kbase = b;
    }
    __syncthreads();
    if (sm.kbase >= 0)
      for (int k = tid; k < kc; k += ADV_BLOCK) fout[sm.kbase + k] = sm.keep[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) kept_min = min(kept_min, (unsigned)__shfl_xor((int)kept_min, o, 64));
  if (lane == 0 && kept_min != 0x7f7fffffu) atomicMin(&c->nf_min_far, kept_min);
}

// One near-far iteration, ONE launch: the advance (relax + far-pile side output), or -- when
// the head moved to a new bucket -- the rebuild of the frontier from the pile.
// KEEP THE LDS OF THIS KERNEL BELOW 64 KB.  Measured (rocprofv3 kernel trace, road stand-in, one launch pair per
// iteration): with 81,920 B of LDS per workgroup the SHORTEST launch of this kernel -- one that only finds `done` set --
// took 13-25 us depending on the box, with 51,712 B it took 2.8 us; 7228 launches per search.  (The 30 KB came from
// a staging array only some policies need and from a per-thread LDS copy of the by-value policy, see sssp_nf_policy::begin.)
__global__ __launch_bounds__(ADV_BLOCK) void sssp_nf_level_kernel(pipe_args a, sssp_nf_args nf, sssp_nf_policy pol,
                                                                  uint32_t xcc_mask) {
  // one of three bodies runs per launch: their LDS is overlaid
#ifdef GRX_NF_CARRY
  using mid_policy = sssp_nf_mid_policy;
#else
  using mid_policy = sssp_nf_policy;
#endif
  constexpr size_t LDS_MID = sizeof(mid_smem<mid_policy>) > sizeof(advance_smem<sssp_nf_policy>) ? sizeof(mid_smem<mid_policy>)
                                                                                               : sizeof(advance_smem<sssp_nf_policy>);
  constexpr size_t LDS_BYTES = LDS_MID > sizeof(split_smem) ? LDS_MID : sizeof(split_smem);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  mid_smem<mid_policy>& msm = *reinterpret_cast<mid_smem<mid_policy>*>(lds_raw);
  advance_smem<sssp_nf_policy>& sm = *reinterpret_cast<advance_smem<sssp_nf_policy>*>(lds_raw);
  split_smem& ssm = *reinterpret_cast<split_smem*>(lds_raw);
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);  // one batch of loads, with nf_split
  const int split = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&c->nf_split, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  if (h.done) return;
  if (split) {
    sssp_split_body(a, nf, pol.dist, ssm);
    return;
  }
  pol.begin(c);
#ifdef GRX_ADVANCE_FIRST
  if (h.mode != 3) {
    advance_block<sssp_nf_policy, false>(a, c, pol, sm, h.level & 1, blockIdx.x, gridDim.x, h.total_chunks,
                                         a.chunk_tile);
    return;
  }
  {
    mid_policy mp;
    static_cast<sssp_nf_policy&>(mp) = pol;
    mid_levels_run(a, c, mp, msm, h, xcc_mask);  // many iterations inside the current bucket, in this one launch
  }
#else
  if (h.mode == 3) {  // many iterations -- and, the far pile permitting, many buckets -- in this one launch
    mid_policy mp;
    static_cast<sssp_nf_policy&>(mp) = pol;
    mid_levels_run(a, c, mp, msm, h, xcc_mask);
    return;
  }
  advance_block<sssp_nf_policy, false>(a, c, pol, sm, h.level & 1, blockIdx.x, gridDim.x, h.total_chunks,
                                       a.chunk_tile);
#endif
}

// Head of a plain (label-correcting) level, ONE launch of one workgroup: as many tiny levels
// as there are (tiny_levels_body), then the bookkeeping + chunk map of the next regular level.
__global__ __launch_bounds__(PLAN_BLOCK) void sssp_head_kernel(pipe_args a, sssp_policy pol, long long n_edges,
                                                               int mid_v, int mid_e, int allow_tiny, bin_args bn, int seq) {
  __shared__ tiny_smem<sssp_policy> tsm;
  __shared__ unsigned long long s_esum[2];
  __shared__ int s_wave[PLAN_BLOCK / 64 + 1];
  static_assert(TINY_THREADS == PLAN_BLOCK, "head kernel runs both bodies");
  const ctrl_head h0 = load_ctrl_head(a.ctrl);
  const int t = allow_tiny ? tiny_levels_body(a, pol, 0, n_edges, tsm, h0) : 0;
  if (t == 1) return;
  const ctrl_head h = t == 2 ? load_ctrl_head(a.ctrl) : h0;
  if (threadIdx.x < 2) s_esum[threadIdx.x] = 0ull;
  __syncthreads();
  plan_in in;
  in.done = h.done;
  in.level = h.level + 1;
  in.nt = h.nt(in.level & 1);
  in.mode = 0;
  in.R = 0;
  in.T = 0;
  in.mid_v = mid_v;  // frontiers of a few thousand vertices: many levels per launch (grx_mid.hpp)
  in.mid_e = mid_e;
  in.mid_tile_e = bn.mid_tile_e;  // ... unless a tile of the frontier is too heavy for one workgroup of that body (plan_in)
  // fat levels of a weighted search on a dense graph: binned relaxation (mode 2, grx_relax.hpp)
  in.bin_min = bn.min_edges;
  in.bin_fill = bn.fill;
  in.bin_queue = bn.queue;
  in.bin_nb = bn.nb;
  in.bin_pad = BIN_PAD;
  in.bin_allowed = bn.allowed;
  in.seq = seq;
  plan_body<PLAN_BLOCK>(a, a.ctrl, 0, s_wave, s_esum, in);
}

// One plain (label-correcting) level, or -- ctrl.mode 3 -- many mid-size levels in this one launch.
__global__ __launch_bounds__(ADV_BLOCK) void sssp_level_kernel(pipe_args a, sssp_policy pol, uint32_t xcc_mask) {
  __shared__ mid_smem<sssp_policy> sm;
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);
  if (h.done || h.mode == 2) return;  // mode 2: this level is the two kernels below
  if (h.mode == 3) {
    mid_levels_run(a, c, pol, sm, h, xcc_mask);
    return;
  }
  pol.begin(c);
  advance_block<sssp_policy, false>(a, c, pol, sm.adv, h.level & 1, blockIdx.x, gridDim.x, h.total_chunks, a.chunk_tile);
}

// A fat level of the plain schedule as a binned relaxation (grx_relax.hpp): scatter, then sweep.  No-ops unless the head chose
// mode 2.  One workgroup of 1024 threads per CU each (96 KB / 108 KB of LDS).
template <bool UNI>
__global__ __launch_bounds__(SC2_BLOCK) void sssp_rscatter_kernel(pipe_args a, bin_args bn) {
  __shared__ __attribute__((aligned(16))) bin_scatter2_val_smem sm;
  const level_head h = load_level_head(a.ctrl);
  if (h.done || h.mode != 2) return;
  bin_scatter2_block<false, true, true, bin_scatter2_val_smem, UNI>(a, bn, sm, h.level & 1, h.total_chunks, a.chunk_tile);
}

__global__ __launch_bounds__(RB_BLOCK) void sssp_rsweep_kernel(pipe_args a, bin_args bn) {
  __shared__ __attribute__((aligned(16))) relax_sweep_smem sm;
  ctrl_t* c = a.ctrl;
  const level_head h = load_level_head(c);
  if (h.done || h.mode != 2) return;
  relax_sweep_block(a, bn, c, h.level, sm, h.level & 1);
}

// out[0] = sum of weights; bits[0] / bits[1] = min / max weight as ordered uints (w >= 0)
__global__ void weight_sum_kernel(const float* w, int64_t n, double* out, unsigned* bits) {
  double acc = 0.0;
  unsigned lo = 0xffffffffu, hi = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = w[i];
    acc += (double)x;
    const unsigned u = x >= 0.0f ? __float_as_uint(x) : 0u;
    lo = min(lo, u);
    hi = max(hi, u);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_xor(acc, o, 64);
    lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
    hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
  }
  if (dev::lane_id() == 0) {
    atomicAdd(out, acc);
    atomicMin(&bits[0], lo);
    atomicMax(&bits[1], hi);
  }
}

// Uniform weights: the search is a BFS and a vertex of depth k carries the label fl(...fl(fl(0 + w) + w)... + w), k additions
// (the sum every shortest path gives the reference's relaxation, sssp.hxx:121-123).  Depths -> distances, in place (the
// caller's float buffer held the int32 depths of the BFS engine).  table == nullptr: w is exactly 1.0, the k-fold sum is
// min(k, 2^24) (16777216 + 1 rounds back to 16777216: the additions stall there, as the reference's do).
// A depth beyond the table means the search under-reported its depth: the pass then raises mailbox word 10 (the host turns it
// into an error) instead of returning the last entry's distance for it.
__global__ void sssp_depth_to_dist_kernel(float* dist, int64_t n, const float* table, int32_t table_n, int32_t* mailbox) {
  int32_t* as_int = reinterpret_cast<int32_t*>(dist);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t k = as_int[i];
    float d = FLT_MAX;  // unreached (sssp.hxx:72-73)
    if (k != INT_MAX) {
      if (table) {
        if (k < 0 || k >= table_n) mailbox[10] = 3;
        d = table[k < 0 ? 0 : (k < table_n ? k : table_n - 1)];
      } else {
        d = (float)(k < (1 << 24) ? k : (1 << 24));
      }
    }
    dist[i] = d;
  }
}

}  // namespace grx

using namespace grx;

// Sum / min / max of the edge weights, once per graph handle (near-far bucket width; graphs whose
// weights are all 1.0 -- what the reference loader makes of a pattern .mtx, io/matrix_market.hxx:170-171
// -- need no weight stream at all).
grx_status_t grx::graph_weight_stats(grx_context_t ctx, grx_graph_t g) {
  std::lock_guard<std::recursive_mutex> lk(g->prep_mu);
  if (g->weight_sum >= 0.0 || !g->w || g->E <= 0) return GRX_SUCCESS;
  GRX_HIP(ctx->misc.reserve(64));
  double* d_sum = reinterpret_cast<double*>(ctx->misc.as<unsigned char>());
  unsigned* d_bits = reinterpret_cast<unsigned*>(d_sum + 1);
  const unsigned init[4] = {0u, 0u, 0xffffffffu, 0u};  // sum = 0.0, min, max
  GRX_HIP(hipMemcpyAsync(d_sum, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(weight_sum_kernel, dim3(1024), dim3(256), 0, ctx->stream, g->w, (int64_t)g->E, d_sum, d_bits);
  unsigned char h[16];
  GRX_HIP(hipMemcpyAsync(h, d_sum, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  unsigned lohi[2];
  memcpy(lohi, h + 8, sizeof(lohi));
  g->uniform_weights = lohi[0] == lohi[1];
  memcpy(&g->weight_min, &lohi[0], sizeof(float));
  memcpy(&g->weight_max, &lohi[1], sizeof(float));
  memcpy(&g->weight_sum, h, sizeof(double));  // (last: weight_sum >= 0 is what says "the statistics are there")
  return GRX_SUCCESS;
}

static int sssp_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// Per-graph static part of the binned relaxation (cached in the handle): bins = runs of granules of about equal numbers of
// in-edges, none wider than RB_WIDTH vertices, at most RB_MAX_BINS of them; capacities = the in-edges of the range (an edge is
// relaxed at most once per level).  Not applicable (state 2) to graphs beyond RB_MAX_BINS * RB_WIDTH = 16.7 M vertices.
static grx_status_t graph_build_relax_bins(grx_context_t ctx, grx_graph_t g) {
  std::lock_guard<std::recursive_mutex> lk(g->prep_mu);
  if (g->rb_state != 0) return GRX_SUCCESS;
  lazy_state state(&g->rb_state);  // an error return leaves it at 0 (the next search tries again), done(2) = not applicable
  prep_timer tm("sssp: relax-bin table (granule counts + cut)", ctx->stream);
  if (g->V <= 0 || g->E <= 0) return state.done(2);
  int gshift = BIN_GSHIFT_MIN;
  while (gshift < 31 && (((long long)g->V + (1ll << gshift) - 1) >> gshift) > BIN_GRAN_MAX) ++gshift;
  if (gshift > RB_SHIFT) return state.done(2);
  const int n_gran = (int)(((long long)g->V + (1ll << gshift) - 1) >> gshift);
  const int max_width = 1 << (RB_SHIFT - gshift);  // granules per bin
  if ((n_gran + max_width - 1) / max_width > RB_MAX_BINS) return state.done(2);
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
  if (total != (long long)g->E) return state.done(2);  // a column index outside [0, V): the relax-per-edge path reports it as before
  std::vector<int> first;
  const int want_bins = std::max(1, std::min(RB_MAX_BINS, sssp_env_int("GRX_RBIN_BINS", 448)));  // (tuning aid; read when the table is built)
  long long target = (total + want_bins - 1) / want_bins;
  // UNIFORM bins (round 5, as for the BFS: grx_bfs.hip graph_build_bins): the aligned 16384-vertex ranges -- bin = id >> 14, the
  // entry is the id, no granule table in the scatter.  Measured on the two dense stand-ins (profiles/r5_c21_relax_uniform_bins.txt):
  // LJ' (14 edges per vertex) 2.995 -> 2.760 ms per search (scatter 1830 -> 1563 us, sweep 851 -> 989), kron' (87) 5.29 -> 5.41 --
  // the value scatter runs one workgroup per CU and feels every instruction; on the kron stand-in the hub ranges' parts cost the
  // sweep more than the scatter gains.  So: uniform below 32 edges per vertex (the opposite of the BFS rule, whose sweep pays per
  // discovered vertex, not per changed label).  GRX_RBIN_UNIFORM=1 | 0 forces one.
  int ushift = 0;
  {
    const int uni_env = sssp_env_int("GRX_RBIN_UNIFORM", -1);
    const long long nb_u = ((long long)g->V + RB_WIDTH - 1) >> RB_SHIFT;
    if ((uni_env > 0 || (uni_env < 0 && (long long)g->E < 32ll * g->V)) && RB_SHIFT >= gshift && nb_u >= 48 && nb_u <= RB_MAX_BINS) {
      ushift = RB_SHIFT;
      for (int i = 0; i < n_gran; i += 1 << (RB_SHIFT - gshift)) first.push_back(i);
    }
  }
  for (int attempt = 0; attempt < 64 && ushift == 0; ++attempt) {
    first.clear();
    long long acc = 0;
    int width = 0;
    for (int i = 0; i < n_gran; ++i) {
      if (width == 0) first.push_back(i);
      acc += cnt[(size_t)i];
      ++width;
      if (acc >= target || width == max_width) { acc = 0; width = 0; }
    }
    if ((int)first.size() <= RB_MAX_BINS) break;
    target += target / 4 + 1;
  }
  const int nb = (int)first.size();
  if (nb < 1 || nb > RB_MAX_BINS) return state.done(2);
  first.push_back(n_gran);
  std::vector<int32_t> offv0(2 * ((size_t)RB_MAX_BINS + 1), 0);
  int32_t* off = offv0.data();
  int32_t* v0 = offv0.data() + RB_MAX_BINS + 1;
  std::vector<unsigned short> g2b16((size_t)BIN_GRAN_MAX, 0);
  long long acc = 0;
  for (int b = 0; b <= RB_MAX_BINS; ++b) {
    off[b] = (int32_t)acc;
    if (b < nb) {
      for (int i = first[(size_t)b]; i < first[(size_t)b + 1]; ++i) {
        acc += cnt[(size_t)i];
        g2b16[(size_t)i] = (unsigned short)(b | ((i - first[(size_t)b]) << 10));
      }
      v0[b] = (int32_t)((long long)first[(size_t)b] << gshift);
    } else {
      v0[b] = (int32_t)((long long)n_gran << gshift);
    }
  }
  static_assert(RB_MAX_BINS <= 1024 && (1 << (RB_SHIFT - BIN_GSHIFT_MIN)) <= 64, "bin | granule index fit 16 bits");
  dev_scratch d_off, d_tab;  // adopted by the handle once everything has arrived
  GRX_HIP(d_off.alloc(offv0.size() * sizeof(int32_t)));
  GRX_HIP(d_tab.alloc((size_t)BIN_GRAN_MAX * sizeof(unsigned short)));
  GRX_HIP(hipMemcpyAsync(d_off.p, offv0.data(), offv0.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  GRX_HIP(hipMemcpyAsync(d_tab.p, g2b16.data(), (size_t)BIN_GRAN_MAX * sizeof(unsigned short), hipMemcpyHostToDevice, s));
  GRX_HIP(hipStreamSynchronize(s));
  g->rb_off = reinterpret_cast<int32_t*>(d_off.release());
  g->rb_g2b16 = reinterpret_cast<unsigned short*>(d_tab.release());
  g->rb_uniform = ushift;
  g->rb_shift = gshift;
  g->rb_ngran = n_gran;
  g->rb_nb = nb;
  return state.done(1);
}

static grx_status_t run_sssp(grx_context_t ctx, grx_graph_t g, int32_t src, const grx_options_t& opt,
                             float* d_dist, bool near_far, float delta, float* elapsed_ms, bool* overflow) {
  pipe_args a;
  grx_status_t st = pipeline_prepare(ctx, g, &a);
  if (st != GRX_SUCCESS) return st;
  // all weights exactly 1.0: nd = d + 1.0f needs no weight stream (4 bytes per relaxed edge less)
  const float* w_eff = graph_unit_weights(g) ? nullptr : g->w;
  GRX_HIP(ctx->labels.reserve((size_t)g->V * sizeof(int32_t)));
  int32_t* stamp = ctx->labels.as<int32_t>();
  hipStream_t s = ctx->stream;
  sssp_nf_args nf{};
  if (near_far) {
    const size_t cap = std::max<size_t>((size_t)2 * (size_t)g->E, (size_t)1 << 20);
    for (auto& b : ctx->far) GRX_HIP(b.reserve(cap * sizeof(int32_t)));
    nf.far[0] = ctx->far[0].as<int32_t>();
    nf.far[1] = ctx->far[1].as<int32_t>();
    nf.capacity = (int32_t)std::min<size_t>(cap, (size_t)0x7fffffff);
  }

  // problem.init()/reset(), outside the timed region as in the reference
  GRX_HIP(fill_f32(s, d_dist, FLT_MAX, g->V));
  GRX_HIP(fill_i32(s, stamp, -1, g->V));

  GRX_HIP(hipEventRecord(ctx->ev_begin, s));
  hipLaunchKernelGGL(sssp_init_kernel, dim3(1), dim3(TILE), 0, s, a, d_dist, src, delta);

  const int grid = advance_grid_for(ctx, g);
  const char* mid_env = getenv("GRX_MID");
  // (the L2-local claim of the mid-level body orders tentative distances by their bit patterns: non-negative only)
  const char* strict_env = getenv("GRX_LB_STRICT");
  const bool strict_mp = ((opt.engine_flags & GRX_FLAG_LB_STRICT) != 0 || (strict_env && *strict_env == '1')) &&
                         (opt.advance_load_balance == GRX_LB_MERGE_PATH || opt.advance_load_balance == GRX_LB_MERGE_PATH_V2);
  const bool mid_on = !strict_mp && !(mid_env && *mid_env == '0') && (!w_eff || g->weight_min > 0.0f);
  const char* mv = getenv("GRX_MID_V");
  const char* me = getenv("GRX_MID_E");
  const int mid_v = mid_on ? ((mv && atoi(mv) > 0) ? atoi(mv) : MID_ENTER_V) : 0;
  const int mid_e = mid_on ? ((me && atoi(me) > 0) ? atoi(me) : MID_ENTER_E) : 0;
  // Binned relaxation of the fat levels (grx_relax.hpp): plain schedule, non-negative weights, dense graphs
  bin_args rb{};
  rb.mid_tile_e = sssp_env_int("GRX_MID_TILE_E", MID_TILE_E);
  rb.xcc_mask = ctx->xcc_mask;
  rb.n_xcd = ctx->n_xcd;
  int grid_rscatter = 0, grid_rsweep = 0;
  const bool want_bins = !near_far && w_eff && g->weight_min > 0.0f && !strict_mp && opt.max_iterations == 0 &&
                         !(opt.engine_flags & (GRX_FLAG_SSSP_NO_BINS | GRX_FLAG_UNFUSED | GRX_FLAG_SYNC_EACH_LEVEL)) &&
                         sssp_env_int("GRX_SSSP_BINS", 1) != 0 &&
                         (long long)g->E >= (long long)sssp_env_int("GRX_RBIN_MIN_GRAPH_EDGES", 1 << 22);
  if (want_bins) {
    st = graph_build_relax_bins(ctx, g);
    if (st != GRX_SUCCESS && !build_failed_softly(st)) return st;
  }
  bool use_rbins = want_bins && st == GRX_SUCCESS && g->rb_state == 1;
  if (use_rbins) {
    // per-SEARCH scratch, owned by the context: 6 bytes per edge.  When it cannot be had the search runs on the relax-per-edge
    // levels -- slower, same result -- instead of failing.
    const size_t fill_bytes = ((size_t)RB_MAX_BINS + 16) * BIN_PAD * sizeof(int32_t);
    if (ctx->rbins[0].reserve(((size_t)g->E + 16) * sizeof(unsigned short)) != hipSuccess ||
        ctx->rbins[1].reserve(((size_t)g->E + 16) * sizeof(unsigned)) != hipSuccess ||
        ctx->rbins[2].reserve(fill_bytes) != hipSuccess) {
      (void)hipGetLastError();
      for (auto& b : ctx->rbins) b.release();
      use_rbins = false;
    }
  }
  if (use_rbins) {
    rb.bins = ctx->rbins[0].as<int32_t>();
    rb.rval = ctx->rbins[1].as<unsigned>();
    rb.fill = ctx->rbins[2].as<int32_t>();
    rb.queue = rb.fill + (size_t)RB_MAX_BINS * BIN_PAD;
    rb.off = g->rb_off;
    rb.v0 = g->rb_off + RB_MAX_BINS + 1;
    rb.g2b16 = g->rb_g2b16;
    rb.gshift = g->rb_shift;
    rb.n_gran = g->rb_ngran;
    rb.nb = g->rb_nb;
    rb.uniform = g->rb_uniform;
    rb.local_ids = 1;
    rb.entry16 = 1;
    // sub-counters per bin in the scatter's LDS histogram (grx_bin.hpp: a hot bin's ranking atomics serialise on one word): as
    // many as the 1024 counters allow
    rb.sub_shift = sssp_env_int("GRX_RBIN_SUB", 1) == 0 ? 0 : (rb.nb <= RB_MAX_BINS / 4 ? 2 : (rb.nb <= RB_MAX_BINS / 2 ? 1 : 0));
    rb.rdist = d_dist;
    rb.rw = w_eff;
    rb.rstamp = stamp;
    rb.min_edges = (long long)sssp_env_int("GRX_RBIN_MIN_EDGES", 1 << 20);
    if (rb.min_edges < 1) rb.min_edges = 1;
    grid_rscatter = ctx->num_cus;
    grid_rsweep = ctx->num_cus * sssp_env_int("GRX_RBIN_SWEEP_WG_PER_CU", 1);
    rb.static_units = (ctx->sc2_static || grid_rscatter < 4 * ctx->n_xcd || sssp_env_int("GRX_SC2_STATIC", 0) != 0) ? 1 : 0;
    rb.fault_xcd = ctx->sc2_static ? 0 : sssp_env_int("GRX_SC2_FAULT_XCD", 0);
    // parts: a bin is cut when it holds more than 1 / GRX_RBIN_PARTS of the level's entries.  Few parts: the parts of a bin pay
    // two device-scope atomics per changed vertex each (a level of 16.7 M entries on the LJ stand-in: 122 us with most bins in two
    // parts, ~60 us with one), many: the one workgroup of a hub range is the tail of the launch
    rb.sweep_items = rb.nb + sssp_env_int("GRX_RBIN_PARTS", ctx->num_cus);
  }
  const uint32_t rhint0 = (use_rbins && sssp_env_int("GRX_BIN_HINT", 1) != 0) ? g->rb_hint.load(std::memory_order_relaxed) : 0u;
  const uint32_t rbin_groups = rhint0 ? (rhint0 | (rhint0 << 1) | (rhint0 >> 1)) : ~0u;
  ctx->levels.clear();
  hipError_t launch_err = hipSuccess;
  // GRX_FLAG_PROFILE: one record per iteration -- head / level kernel times from events on this
  // stream (run_levels synchronises after every group in that mode), frontier size and relaxed
  // edges from the control block's running totals
  const bool profile = (opt.engine_flags & GRX_FLAG_PROFILE) != 0;
  hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
  if (profile) for (auto& e : pe) GRX_HIP(hipEventCreate(&e));
  int64_t prof_v = 0, prof_e = 0;
  auto group = [&](hipStream_t stream, auto&& head, auto&& level) {
    if (profile) (void)hipEventRecord(pe[0], stream);
    head();
    if (profile) (void)hipEventRecord(pe[1], stream);
    level();
    if (profile) {
      (void)hipEventRecord(pe[2], stream);
      (void)hipEventSynchronize(pe[2]);
      level_rec r{};
      (void)hipEventElapsedTime(&r.other_ms, pe[0], pe[1]);
      (void)hipEventElapsedTime(&r.advance_ms, pe[1], pe[2]);
      ctx->levels.push_back(r);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) launch_err = e;
  };
  auto after = [&](const ctrl_t& h) {
    if (!profile || ctx->levels.empty()) return;
    level_rec& r = ctx->levels.back();
    r.frontier_size = h.vertices_visited - prof_v;
    r.edges = h.edges_visited - prof_e;
    // near-far: 2 = this iteration only pulled a bucket out of the far pile; plain: the level's mode (2 = binned relaxation,
    // 3 = many levels in this launch)
    r.bottom_up = near_far ? 2 * h.nf_split : h.mode;
    r.bu_open = h.level;    // (SSSP records: the level the search stands at behind this group -- many levels per launch show as jumps --
    r.bu_probes = h.mode;   // and the body its level kernel ran: 0 relax-per-edge advance, 2 binned relaxation, 3 many levels)
    prof_v = h.vertices_visited;
    prof_e = h.edges_visited;
    // the group that only detected the end carries no work; one that ran the last iterations itself (many per
    // launch, grx_mid.hpp) and found the end is a record like any other
    if (h.done && r.frontier_size == 0 && r.edges == 0) ctx->levels.pop_back();
  };
  if (near_far) {
    sssp_nf_policy pol{d_dist, stamp, w_eff ? w_eff : g->w, nf, 0, 0.0f, nullptr, nullptr, nullptr, 0, 0, 0.0f};
    a.mid_refill_max = sssp_env_int("GRX_NF_FOLD", 65536);  // (0: every bucket change through the head kernel, as before round 6)
    st = run_levels(ctx, opt, [&](hipStream_t stream, int) {
      group(stream,
            [&] { hipLaunchKernelGGL(sssp_nf_head_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, a, nf, mid_v, mid_e); },
            [&] { hipLaunchKernelGGL(sssp_nf_level_kernel, dim3(grid), dim3(ADV_BLOCK), 0, stream, a, nf, pol, ctx->xcc_mask); });
    }, after);
  } else {
    sssp_policy pol{d_dist, stamp, w_eff, 0, 0, (!g->w || g->uniform_weights) ? 1 : 0};
    // Dense graphs (a few dozen fat levels, one per group): the first blind batch is as many groups as the previous search on
    // the graph ran levels (+ the one that finds the frontier empty), instead of 4 + 8 + 16 with a host round trip after each
    // and up to half of the last batch wasted (LJ stand-in, U{1..1000}: 18 levels -> 28 groups, 3 round trips).
    // GRX_GROUP_HINT=0: off
    const bool dense_plain = g->V > 0 && (long long)g->E >= 8ll * g->V && opt.max_iterations == 0;
    const int hinted = (dense_plain && sssp_env_int("GRX_GROUP_HINT", 1) != 0) ? g->group_hint[2].load(std::memory_order_relaxed) : 0;
    const int first_batch = hinted > 0 ? std::min(std::max(hinted, 4), 64) : 4;
    st = run_levels(ctx, opt, [&](hipStream_t stream, int seq) {
      // (as for the BFS: the two kernels of a binned level ride only in the groups where the previous search on the graph
      // met a fat level, one group of slack either side; a fat level elsewhere runs on the relax-per-edge advance)
      const bool bins_here = use_rbins && (seq >= 32 || ((rbin_groups >> seq) & 1u) != 0u);
      rb.allowed = bins_here ? 1 : 0;
      group(stream,
            [&] { hipLaunchKernelGGL(sssp_head_kernel, dim3(1), dim3(PLAN_BLOCK), 0, stream, a, pol, (long long)g->E, mid_v, mid_e, strict_mp ? 0 : 1, rb, seq); },
            [&] {
              hipLaunchKernelGGL(sssp_level_kernel, dim3(grid), dim3(ADV_BLOCK), 0, stream, a, pol, ctx->xcc_mask);
              if (bins_here) {
                if (rb.uniform) hipLaunchKernelGGL(sssp_rscatter_kernel<true>, dim3(grid_rscatter), dim3(SC2_BLOCK), 0, stream, a, rb);
                else hipLaunchKernelGGL(sssp_rscatter_kernel<false>, dim3(grid_rscatter), dim3(SC2_BLOCK), 0, stream, a, rb);
                hipLaunchKernelGGL(sssp_rsweep_kernel, dim3(grid_rsweep), dim3(RB_BLOCK), 0, stream, a, rb);
              }
            });
    }, after, first_batch, /*pace_depth=*/0, /*fast_return=*/false, nullptr, 0, nullptr, /*batch_after_first=*/hinted > 0 ? 4 : 0);
    if (st == GRX_SUCCESS && dense_plain && ctx->h_ctrl->done && ctx->h_ctrl->mid_err == 0)
      g->group_hint[2].store(ctx->h_ctrl->level + 1, std::memory_order_relaxed);
  }
  if (profile) for (auto& e : pe) (void)hipEventDestroy(e);
  if (st != GRX_SUCCESS) return st;
  if (launch_err != hipSuccess) return fail(GRX_ERROR_HIP, hipGetErrorString(launch_err));
  if (ctx->h_ctrl->mid_err != 0 || ctx->h_mailbox[10] != 0) {
    const int code = ctx->h_mailbox[10] != 0 ? (int)ctx->h_mailbox[10] : (int)ctx->h_ctrl->mid_err;
    ctx->h_mailbox[10] = 0;
    if (code == 2) {
      // the bins of a level did not receive exactly its out-edges (grx_bin.hpp, safety net of the scatter's per-XCD queues):
      // nothing wrong was returned -- the search stopped there.  Repeat it with statically strided units, and keep that mode.
      GRX_HIP(hipStreamSynchronize(s));
      GRX_HIP(hipMemsetAsync(&ctx->d_ctrl->mid_err, 0, sizeof(int32_t), s));
      if (!ctx->sc2_static) {
        ctx->sc2_static = true;
        return run_sssp(ctx, g, src, opt, d_dist, near_far, delta, elapsed_ms, overflow);
      }
      return fail(GRX_ERROR_HIP, "grx_sssp: the binned scatter did not cover the level's edges (grx_bin.hpp)");
    }
    return fail(GRX_ERROR_HIP, "grx_sssp: a device-side barrier timed out (grx_mid.hpp)");
  }
  if (use_rbins) g->rb_hint.fetch_or((uint32_t)ctx->h_ctrl->bin_want, std::memory_order_relaxed);

  GRX_HIP(hipEventRecord(ctx->ev_end, s));
  GRX_HIP(hipEventSynchronize(ctx->ev_end));
  float ms = 0;
  GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  ctx->stats.edges_visited = ctx->h_ctrl->edges_visited;
  ctx->stats.vertices_visited = ctx->h_ctrl->vertices_visited;
  ctx->stats.search_depth = ctx->h_ctrl->level;
  ctx->stats.elapsed_ms = ms;
  ctx->stats.n_levels_recorded = (int32_t)ctx->levels.size();
  ctx->stats.reserved = near_far ? (float)ctx->h_ctrl->nf_phases : 0.0f;
  if (overflow) *overflow = near_far && ctx->h_ctrl->nf_overflow != 0;
  if (elapsed_ms) *elapsed_ms = ms;
  return GRX_SUCCESS;
}

static grx_status_t sssp_uniform_as_bfs(grx_context_t ctx, grx_graph_t g, int32_t src, const grx_options_t& opt,
                                        float* d_dist, float w, float* elapsed_ms) {
  grx_options_t bo = opt;
  // only the bits the BFS engine knows: the SSSP schedule bits 0x10 .. 0x80 mean nothing to it, and bits 0x100 .. 0x700 select
  // its kernel VARIANT
  bo.engine_flags &= (GRX_FLAG_UNFUSED | GRX_FLAG_PROFILE | GRX_FLAG_SYNC_EACH_LEVEL | GRX_FLAG_ASYNC_RETURN | GRX_FLAG_LB_STRICT |
                      GRX_FLAG_NO_BLOCK_ASYNC);
  // The search returns the moment the device has published its end (paced searches: depth and counters come with the flag
  // through the mailbox, which is all the table of a non-unit weight needs); the conversion pass is queued behind it at once
  // and THIS call blocks once, on the pass -- not twice (0.517 -> 0.49 ms on the LJ stand-in).  GRX_SSSP_BFS_ASYNC=0: the
  // search blocks as before.
  {
    const char* ba = getenv("GRX_SSSP_BFS_ASYNC");
    if (ba && *ba == '0') bo.engine_flags &= ~GRX_FLAG_ASYNC_RETURN;
    else bo.engine_flags |= GRX_FLAG_ASYNC_RETURN;
  }
  float bfs_ms = 0.0f;
  grx_status_t st = grx_bfs(ctx, g, src, &bo, reinterpret_cast<int32_t*>(d_dist), nullptr, &bfs_ms);
  if (st != GRX_SUCCESS) return st;
  hipStream_t s = ctx->stream;
  const int32_t depth = ctx->stats.search_depth;
  const float* d_table = nullptr;
  int32_t table_n = 0;
  if (w != 1.0f) {
    // t[k] = fl(t[k-1] + w), capped at FLT_MAX: a tentative distance that is not < FLT_MAX never replaces the initial label
    // (sssp.hxx:121-126), so such a vertex keeps FLT_MAX
    std::vector<float> t((size_t)std::max(depth, 0) + 2);
    t[0] = 0.0f;
    for (size_t k = 1; k < t.size(); ++k) {
      const float nd = t[k - 1] + w;
      t[k] = nd < FLT_MAX ? nd : FLT_MAX;
    }
    GRX_HIP(ctx->misc.reserve(t.size() * sizeof(float) + 64));
    float* dt = reinterpret_cast<float*>(ctx->misc.as<unsigned char>() + 64);
    GRX_HIP(hipMemcpyAsync(dt, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, s));
    GRX_HIP(hipStreamSynchronize(s));  // t is a local
    d_table = dt;
    table_n = (int32_t)t.size();
  }
  hipEvent_t e0 = ctx->ev_begin, e1 = ctx->ev_end;
  GRX_HIP(hipEventRecord(e0, s));
  hipLaunchKernelGGL(sssp_depth_to_dist_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, s, d_dist, (int64_t)g->V, d_table, table_n,
                     ctx->d_mailbox);
  GRX_HIP(hipEventRecord(e1, s));
  GRX_HIP(hipEventSynchronize(e1));
  if (ctx->h_mailbox[10] == 3) {
    ctx->h_mailbox[10] = 0;
    return fail(GRX_ERROR_HIP, "grx_sssp: a depth beyond the reported search depth (uniform weights on the BFS engine)");
  }
  float conv_ms = 0.0f;
  GRX_HIP(hipEventElapsedTime(&conv_ms, e0, e1));
  ctx->stats.elapsed_ms = bfs_ms + conv_ms;  // enact() scope: the search and the pass that writes the distances
  if (elapsed_ms) *elapsed_ms = ctx->stats.elapsed_ms;
  return GRX_SUCCESS;
}

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

  // bucket width: 64 x (mean weight) / (mean degree)  (2 x Davidson et al.'s constant, see below); unit-weight
  // graphs (values == NULL) are already level-synchronous => plain schedule
  bool near_far = g->w != nullptr && g->E > 0 && !(opt.engine_flags & GRX_FLAG_SSSP_PLAIN);
  float delta = FLT_MAX;
  {
    grx_status_t wst = graph_weight_stats(ctx, g);
    if (wst != GRX_SUCCESS) return wst;
  }
  // All weights equal and non-negative -- what the reference loader makes of every pattern .mtx (io/matrix_market.hxx:170-171:
  // soc-LiveJournal1, kron_g500-logn21, road_usa, soc-twitter-2010 as distributed) -- : the search IS a breadth-first
  // search and a vertex of depth k gets the k-fold fp32 sum of w.  It runs on the BFS engine (binned fat levels, many
  // mid-size levels per launch, bottom-up levels with advance_direction = optimized) writing int32 depths into the
  // caller's buffer, and one streaming pass turns them into distances.  GRX_FLAG_SSSP_NO_BFS / GRX_SSSP_UNIFORM_BFS=0:
  // the relaxation kernels below, as for any other weights.
  {
    const char* ub = getenv("GRX_SSSP_UNIFORM_BFS");
    const bool all_equal = !g->w || (g->E > 0 && g->uniform_weights && g->weight_min == g->weight_max &&
                                     (g->weight_min > 0.0f || g->weight_sum == 0.0));
    if (all_equal && !(opt.engine_flags & (GRX_FLAG_SSSP_NO_BFS | GRX_FLAG_SSSP_PLAIN | GRX_FLAG_UNFUSED)) &&
        !(ub && *ub == '0'))
      return sssp_uniform_as_bfs(ctx, g, src, opt, d_dist, g->w ? g->weight_min : 1.0f, elapsed_ms);
  }
  // weighted road-like graphs: block-asynchronous relaxation (grx_block.hip) instead of the near-far schedule
  ctx->block_stats = grx_block_stats_t{};
  if (g->w && g->E > 0 && opt.max_iterations == 0 &&
      !(opt.engine_flags & (GRX_FLAG_NO_BLOCK_ASYNC | GRX_FLAG_SSSP_PLAIN | GRX_FLAG_SSSP_NEAR_FAR | GRX_FLAG_UNFUSED |
                            GRX_FLAG_SYNC_EACH_LEVEL | GRX_FLAG_LB_STRICT))) {
    const char* strict = getenv("GRX_LB_STRICT");
    bool use = false;
    if (!(strict && *strict == '1')) {
      grx_status_t bst = blk_prepare(ctx, g, true, &use);
      if (bst != GRX_SUCCESS && !build_failed_softly(bst)) return bst;
      if (bst != GRX_SUCCESS) use = false;
    }
    if (use) return blk_search(ctx, g, src, opt, true, d_dist, elapsed_ms);
  }
  if (near_far) {
    // all weights equal (e.g. a pattern .mtx loaded with 1.0 everywhere): the search is
    // level-synchronous already, nothing is ever re-relaxed
    if (g->uniform_weights) near_far = false;
    const double mean_w = g->weight_sum / (double)g->E;
    const double mean_deg = std::max(1.0, (double)g->E / (double)std::max(1, g->V));
    // GRX_NF_DELTA_SCALE: tuning knob (bucket width multiplier)
    const char* dsc = getenv("GRX_NF_DELTA_SCALE");
    // Rounds 1-5: width 128 x mean weight / mean degree, four times Davidson et al.'s constant.  Measured on the
    // weighted road stand-in (tools/history/ab_sssp_delta.py), width / iterations / relaxations / time:
    //   16: 10417 / 65 M / 192 ms   32: 8603 / 74 M / 164 ms   64: 7463 / 92 M / 150 ms
    //   128: 6749 / 130 M / 143 ms   256: 6313 / 213 M / 144 ms
    // -- an iteration is ~19 us of launch and latency, so fewer, fatter iterations win until the
    // extra relaxations catch up.
    // Round 6: the whole search of a road network is ONE launch now (buckets change inside it, grx_mid.hpp `policy_refills`), a
    // level costs ~8 us + its relaxations, and a bucket change costs no launch: width / iterations / relaxations / time
    //   32: 8587 / 72 M / 81.1 ms   64: 7457 / 87 M / 75.5 ms   96: 7001 / 104 M / 76.7 ms   128: 6742 / 121 M / 79.8 ms
    //   192: 6475 / 158 M / 89.0 ms   256: 6307 / 197 M / 99.7 ms          (profiles/r6_c13_delta.txt)
    const double dlt = 64.0 * mean_w / mean_deg * ((dsc && atof(dsc) > 0.0) ? atof(dsc) : 1.0);
    if (!(mean_w > 0.0) || !std::isfinite(dlt) || dlt <= 0.0) near_far = false;  // zero / negative weights
    // dense, low-diameter graphs finish in a dozen levels: label-correcting wastes little
    // there and the pile handling only costs (measured: LJ stand-in 4.8 ms plain vs 7.2 ms)
    if (mean_deg >= 6.0 && !(opt.engine_flags & GRX_FLAG_SSSP_NEAR_FAR)) near_far = false;
    else delta = (float)dlt;
  }
  bool overflow = false;
  grx_status_t st = run_sssp(ctx, g, src, opt, d_dist, near_far, delta, elapsed_ms, &overflow);
  if (st != GRX_SUCCESS) return st;
  if (overflow)  // far pile exceeded its capacity: distances may be incomplete -- redo plainly
    st = run_sssp(ctx, g, src, opt, d_dist, false, FLT_MAX, elapsed_ms, nullptr);
  return st;
}
