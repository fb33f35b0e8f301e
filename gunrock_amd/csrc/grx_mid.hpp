// grx_mid.hpp -- MANY MID-SIZE LEVELS IN ONE LAUNCH: a few workgroups keep expanding the frontier
// level after level, separated by a device-side barrier, instead of one head + level launch pair per
// level.
//
// What it replaces in the reference: the host loop of enactor_t::enact() (framework/enactor.hxx:274-277)
// with its >= 2 blocking synchronisations per level, for the levels of a high-diameter search.
//
// Why.  On a road network a BFS / SSSP level has a few thousand vertices and ~10 k edges, and there are
// thousands of them.  Round 1 ran such a level as a head kernel (one workgroup: bookkeeping + chunk map)
// plus a level kernel: two launches, ~13 us per level, nearly all of it launch overhead and dependent
// round trips (the LDS-resident tiny-level body covers only frontiers of <= 4096 edges: one CU retires
// about one scattered transaction every two clocks).  Here MID_WGS workgroups of the level kernel stay
// resident and run level after level themselves:
//   * the frontier is a FLAT queue in the same two parity buffers the tile queue uses; workgroup w
//     takes slots w * 256, (w + MID_WGS) * 256, ...: a block of 256 vertices is staged, its degrees
//     scanned in LDS, and the lanes walk consecutive edges exactly as advance_block does (same policy
//     interface, same phased loads / atomics), so hubs are shared by the 256 lanes of the workgroup;
//   * accepted neighbours are compacted in LDS and appended to the next queue with one reservation atomic
//     per 256 (or per level and workgroup);
//   * a level ends at a counter barrier among the resident workgroups (every thread drains its stores, one
//     arrival atomic per workgroup, one lane polls).
//   * ALL OF THEM RUN ON ONE XCD.  The first version let any 32 workgroups stay and talked through agent-scope
//     (memory-side) atomics and sc1 stores: a level cost 11.5 us on the road stand-in -- ten dependent round
//     trips of ~1 us each across the fabric -- against 13 us for the launch pair.  Here only workgroups whose
//     CU reports the HOME XCC id (HW_REG_XCC_ID) register -- a short registration window fixes their number,
//     so nothing depends on how the grid was dispatched -- and everything they share lives in that XCD's L2:
//     plain stores (the L1 writes through), loads that bypass the L1 (sc1: L2-served), and WORKGROUP-scope
//     atomics, which execute in the L2 -- for the barrier, the queue counters and the label claims of the
//     policy alike.  That is coherent because nobody else touches these words during the launch (the rest
//     of the grid leaves at once), and it reaches memory at the end of the kernel like any other store.
// The head kernel chooses (frontier <= MID_ENTER_V vertices and <= MID_ENTER_E out-edges -> ctrl.mode 3);
// the body leaves when the search ends (it publishes `done` like every other kernel) or when the
// frontier outgrows MID_EXIT_V, handing the queue back as tiles with their metadata.  Every spin is
// bounded: a barrier that does not complete raises ctrl.mid_err instead of hanging the device.
#pragma once

// The multi-level bodies and the l2_local claims rely on two gfx950 (CDNA3/4 multi-XCD) facts: s_getreg_b32 0x1814 is
// HW_REG_XCC_ID, and workgroup-scope atomics / sc1 loads are coherent inside ONE XCD's L2.  Any other target would turn
// claims silently wrong, so the device pass refuses to compile for it.
// (gfx950 ONLY: the sweep / relax kernels keep 82-111 KB of static LDS per workgroup -- gfx942 has 64 KB -- and the DPP scans of
// include/gunrock/hip/wave.hxx use gfx950 row controls.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "grx_mid.hpp is written for gfx950 (XCC id register, XCD-local L2 coherence, 160 KB of LDS); build with --offload-arch=gfx950"
#endif


#include "grx_frontier.hpp"

namespace grx {

// The bodies are large; GRX_MID_NOINLINE keeps them out of line (experiment: code layout of the launch-bound kernels)
#ifdef GRX_MID_NOINLINE
#define GRX_MID_FN __device__ __attribute__((noinline))
#else
#define GRX_MID_FN __device__ __forceinline__
#endif

constexpr int MID_WGS = GRX_MID_WGS; // workgroups that stay (32: one per CU of an XCD; the rest of the grid leaves at once)
static_assert(MID_WGS == 32 || MID_WGS == 64, "the exchange reads every workgroup's word with one wave");
constexpr int MID_ENTER_V = 8192;    // a level enters with at most this many frontier vertices ...
constexpr int MID_ENTER_E = 65536;   // ... and out-edges
constexpr int MID_TILE_E = 4096;     // ... none of its tiles (the 256 slots one workgroup stages) holding more than two chunks of them
constexpr int MID_EXIT_V = 131072;   // a frontier beyond this goes back to the regular kernels
constexpr int MID_EXIT_E = 4 * MID_ENTER_E;  // ... and so does one with more out-edges than this (8 chunks per workgroup)
constexpr int MID_SPIN_LIMIT = 1 << 22;
#ifndef GRX_MID_TM_ITEMS
#define GRX_MID_TM_ITEMS 8
#endif
constexpr int MID_TM_ITEMS = GRX_MID_TM_ITEMS;  // a block whose rows are all at most this long is expanded thread-mapped (second build of the chunk)
constexpr int MID_AUX_CAP = MID_EXIT_V + TILE;  // queue entries that carry their row start / degree along
// second version (mid_levels_body2): every workgroup appends to a PRIVATE region of the next queue -- no reservation
// atomic -- and the barrier carries the counts; what does not fit goes to a shared overflow area behind the regions
static_assert(MID_FLAG_WORDS == 2 * MID_WGS, "two sets of flag words");
constexpr int MID_SEG = 16384 * 32 / MID_WGS;
constexpr int MID_SEG_TILES = MID_SEG / TILE;
constexpr int MID_OVF_BASE = MID_WGS * MID_SEG;
constexpr int MID_AUX2_CAP = MID_OVF_BASE;      // int4 {row start, degree, state, -} per entry of the private regions

// optional policy hooks of the second version: `bool carry_state() const` -- the state a vertex is expanded from is
// known when it is claimed (`src_state state_of(int cand)`), so it travels with the queue entry instead of being
// loaded again (one dependent round trip less per level)
template <class Policy, class = void>
struct policy_carries_state : std::false_type {};
template <class Policy>
struct policy_carries_state<Policy, std::void_t<decltype(&Policy::carry_state)>> : std::true_type {};

template <class Policy>
struct mid_smem {
  advance_smem<Policy> adv;
#if defined(GRX_MID_BOTH) || defined(GRX_MID_FIRST_VERSION)
  int out_rs[TILE + CHUNK];   // first version: row start / degree of the staged output vertices (parallel to adv.out)
  int out_deg[TILE + CHUNK];
#endif
  int tcount[ADV_BLOCK];      // entering level: counts of this workgroup's next 256 tiles
  int side[policy_has_side<Policy>::value ? (TILE + CHUNK) : 1];  // staged side-pile entries (near-far SSSP)
  int seg_pre[MID_WGS + 1];   // second version: entries of the private regions before region i
  int side_cnt;
  int side_base;
  int base;
  int n_next;
  int n_ovf;
  int ok;
  int rank;
  int out_edges;      // second version: out-degree sum of the entries this workgroup appended this level
  int ovf;            // ... and whether any of them went to the shared overflow area
  int next_edges;     // ... and (in units of 128, from the exchange words) of the whole next level
};



// optional policy hook: `prepare(src_state, nbr, edge, cand&)` -- what precheck computes WITHOUT the read-only
// probe of the neighbour's label.  Inside one XCD the claim itself is an L2 operation, cheaper than the extra
// dependent round trip of a probe that mostly misses to HBM.
template <class Policy, class = void>
struct policy_has_prepare : std::false_type {};
template <class Policy>
struct policy_has_prepare<Policy, std::void_t<decltype(&Policy::prepare)>> : std::true_type {};

// An empty frontier ends the search -- unless the policy says otherwise (near-far SSSP: the BUCKET is drained and
// the head kernel moves on to the next one): `static constexpr bool drained_is_done = false`.
template <class Policy, class = void>
struct policy_drained_is_done : std::true_type {};
template <class Policy>
struct policy_drained_is_done<Policy, std::void_t<decltype(Policy::drained_is_done)>> : std::bool_constant<Policy::drained_is_done> {};

// optional policy hooks `refill_decide / refill_rebind / refill_entry / refill_class / refill_keep_store` (near-far SSSP, round 6):
// a drained frontier is not the end of the launch -- the next bucket is pulled out of the policy's side pile by the resident
// workgroups (second version of the body only; see sssp_nf_policy)
template <class Policy, class = void>
struct policy_refills : std::false_type {};
template <class Policy>
struct policy_refills<Policy, std::void_t<decltype(&Policy::refill_decide)>> : std::true_type {};

// policies whose claims can be told to execute in the local L2 (see above)
template <class Policy, class = void>
struct policy_has_l2_local : std::false_type {};
template <class Policy>
struct policy_has_l2_local<Policy, std::void_t<decltype(std::declval<Policy&>().l2_local)>> : std::true_type {};

// Barrier among the n_wg resident workgroups.  epoch: barriers passed so far in this launch (uniform).
// Returns false if it timed out.
template <class Policy>
__device__ __forceinline__ bool mid_barrier(ctrl_t* c, int n_wg, int& epoch, mid_smem<Policy>& sm) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores have been acknowledged
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&c->mid_bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // in the home L2
    const int want = n_wg * epoch;
    int spins = 0, ok = 1;
    while (__hip_atomic_load(&c->mid_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > MID_SPIN_LIMIT) { ok = 0; break; }
    }
    sm.ok = ok;
  }
  __syncthreads();
  return sm.ok != 0;
}

// Append sm.adv.out[lo .. lo + k) to the next queue (count word *cnt_out).  Block-wide.
template <class Policy>
__device__ __forceinline__ void mid_flush(int32_t* qout, int2* aux_out, int* cnt_out, mid_smem<Policy>& sm, int lo, int k) {
  if (threadIdx.x == 0) sm.base = __hip_atomic_fetch_add(cnt_out, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  const int base = sm.base;
  for (int i = threadIdx.x; i < k; i += ADV_BLOCK) {
    qout[base + i] = sm.adv.out[lo + i];  // through the L1 into the home L2
    if (base + i < MID_AUX_CAP) aux_out[base + i] = make_int2(sm.out_rs[lo + i], sm.out_deg[lo + i]);
  }
  __syncthreads();
}

// Runs in the level kernel when ctrl.mode == 3.  h: the control block as the kernel read it (level = the level
// to expand, its frontier is the tile queue of that parity).
template <class Policy>
GRX_MID_FN void mid_levels_body(const pipe_args& a, ctrl_t* c, Policy& pol, mid_smem<Policy>& sm,
                                                const level_head& h, uint32_t xcc_mask) {
  constexpr bool SIDE = policy_has_side<Policy>::value;
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  // ---- who takes part: the workgroups on the home XCD that register before the leader closes the window
  const unsigned my_xcc = (unsigned)__builtin_amdgcn_s_getreg(0x1814) & 15u;
  if (my_xcc != (unsigned)__builtin_ctz(xcc_mask ? xcc_mask : 1u)) return;
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(&c->mid_reg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    sm.rank = (old >> 31) ? -1 : (int)(old & 0xffffu);
  }
  __syncthreads();
  const int w = sm.rank;
  if (w < 0) return;  // the window had closed
  if (tid == 0) {
    int g = 0;
    if (w == 0) {
      // leader: wait (a few us at most) for the expected number of registrations, then close
      const int expect = min(MID_WGS, max(1, (int)gridDim.x / max(1, __popc(xcc_mask))));
      for (int i = 0; i < 64; ++i) {
        if ((int)(__hip_atomic_load(&c->mid_reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu) >= expect) break;
        __builtin_amdgcn_s_sleep(8);
      }
      const unsigned old = __hip_atomic_fetch_or(&c->mid_reg, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      g = min((int)(old & 0xffffu), MID_WGS);
      __hip_atomic_store(&c->mid_G, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      int spins = 0;
      while ((g = __hip_atomic_load(&c->mid_G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > MID_SPIN_LIMIT) { g = -1; break; }
      }
    }
    sm.n_next = g;
  }
  __syncthreads();
  const int G = sm.n_next;
  __syncthreads();
  if (G < 0) {
    if (tid == 0) { c->mid_err = 1; c->done = 1; a.mailbox[10] = 1; __threadfence_system(); a.mailbox[0] = 1; }
    return;
  }
  if (w >= G) return;
  if constexpr (policy_has_l2_local<Policy>::value) pol.l2_local = 1;
  advance_smem<Policy>& ad = sm.adv;
  int level = h.level;
  int epoch = 0;
  bool first = true;                       // the entering frontier is a tile queue (slots may be -1)
  int n_in = ((level & 1) ? h.nt1 : h.nt0) * TILE;       // slots to look at
  long long my_edges = 0, my_vertices = 0; // levels AFTER the entering one (the head accounted for that one)
  for (;;) {
    const int p = level & 1;
    const int32_t* qin = a.frontier[p];
    int32_t* qout = a.frontier[p ^ 1];
    const int2* aux_in = reinterpret_cast<const int2*>(a.mid_aux) + (size_t)p * MID_AUX_CAP;
    int2* aux_out = reinterpret_cast<int2*>(a.mid_aux) + (size_t)(p ^ 1) * MID_AUX_CAP;
    int* cnt_out = &c->mid_cnt[(level + 1) % 3];
    // the counter the level after next appends to (last read one level ago)
    if (w == 0 && tid == 0) __hip_atomic_store(&c->mid_cnt[(level + 2) % 3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    pol.set_level(level);
    if (tid == 0) { ad.cnt = 0; sm.side_cnt = 0; }
    __syncthreads();
    // Blocks of 256 slots this workgroup expands: blocks w, w + G, ... of the flat queue; on the entering level the
    // NON-EMPTY tiles among tiles w, w + G, ... (a tile queue is mostly reserved-but-empty tiles after a level of
    // the regular kernels -- 3072 tiles for a few thousand vertices on the LJ stand-in: their counts are fetched
    // 256 at a time, one round trip, and the empty ones cost an LDS read each)
    const int n_blocks = (n_in + TILE - 1) / TILE;
    for (int i0 = 0; w + i0 * G < n_blocks; i0 += first ? ADV_BLOCK : 1) {
      int span = 1;
      if (first) {
        const int t = w + (i0 + tid) * G;
        sm.tcount[tid] = t < n_blocks ? a.tile_count[t] : 0;
        span = ADV_BLOCK;
        __syncthreads();
      }
      for (int j = 0; j < span; ++j) {
      const int blk = w + (i0 + j) * G;
      if (blk >= n_blocks) break;
      int my_count = TILE;
      if (first) {
        my_count = sm.tcount[j];
        if (my_count == 0) continue;  // uniform
      }
      const int base = blk * TILE;
      int v = -1;
      if (first) {
        if (tid < my_count) v = qin[base + tid];  // tiles are front-packed: only the first tile_count slots were written
      } else if (base + tid < n_in) {
        v = __hip_atomic_load(&qin[base + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      int rs = 0, deg = 0;
      typename Policy::src_state st{};
      if (v >= 0) {
        if (!first && base + tid < MID_AUX_CAP) {
          // the producer of this entry fetched its row start / degree while it waited at the barrier anyway
          const long long rd = __hip_atomic_load(reinterpret_cast<const long long*>(&aux_in[base + tid]), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
          rs = (int)(rd & 0xffffffffll);
          deg = (int)(rd >> 32);
        } else {
          rs = a.ro[v];
          deg = a.ro[v + 1] - rs;
        }
        st = pol.load_source(v);
      }
      int tot;
      const int ex = dev::block_exclusive_sum<ADV_BLOCK>(deg, ad.wave, &tot);
      ad.seg[tid] = ex;
      ad.start[tid] = rs;
      ad.src[tid] = v;
      ad.state[tid] = st;
      if (tid == 0) ad.seg[TILE] = tot;
      const int n_valid = __syncthreads_count(v >= 0);
      if (!first) {
        my_edges += tot;
        my_vertices += n_valid;
      }
      for (int a0 = 0; a0 < tot; a0 += CHUNK) {
        const int a_end = min(tot, a0 + CHUNK);
        int e_k[ADV_ITEMS], slot_k[ADV_ITEMS], n_k[ADV_ITEMS], cand_k[ADV_ITEMS];
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) {
          const int atom = a0 + k * ADV_BLOCK + tid;
          int lo = 0;
          if (atom < a_end) {
#pragma unroll
            for (int step = TILE / 2; step >= 1; step >>= 1)
              if (ad.seg[lo + step] <= atom) lo += step;
            e_k[k] = ad.start[lo] + (atom - ad.seg[lo]);
          } else {
            e_k[k] = -1;
          }
          slot_k[k] = lo;
        }
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) n_k[k] = a.ci[e_k[k] >= 0 ? e_k[k] : 0];
        bool pre_k[ADV_ITEMS];
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) {
          const bool ok = e_k[k] >= 0;
          cand_k[k] = 0;
          bool pass;
          if constexpr (policy_has_prepare<Policy>::value)
            pass = pol.prepare(ad.state[slot_k[k]], n_k[k], ok ? e_k[k] : 0, cand_k[k]);
          else
            pass = pol.precheck(ad.state[slot_k[k]], n_k[k], ok ? e_k[k] : 0, cand_k[k]);
          pre_k[k] = pass & ok;
        }
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
        bool keep_k[ADV_ITEMS];
        int nrs_k[ADV_ITEMS], nre_k[ADV_ITEMS], code_k[ADV_ITEMS];
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) {
          code_k[k] = 0;
          if (pre_k[k]) code_k[k] = pol.code(r1_k[k], r2_k[k], n_k[k], cand_k[k]);
          keep_k[k] = code_k[k] == 1;
        }
        // row offsets of the accepted vertices, for the level that expands them: unconditional loads from a
        // clamped index (the others read row 0), all in flight together
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) {
          const int x = keep_k[k] ? n_k[k] : 0;
          nrs_k[k] = a.ro[x];
          nre_k[k] = a.ro[x + 1];
        }
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) {
          const bool keep = keep_k[k];
          const unsigned long long m = dev::ballot(keep);
          if (m) {
            int at = 0;
            if (lane == 0) at = atomicAdd(&ad.cnt, __popcll(m));
            at = dev::wave_bcast0(at);
            if (keep) {
              const int pos = at + dev::mask_rank(m);
              ad.out[pos] = n_k[k];
              sm.out_rs[pos] = nrs_k[k];
              sm.out_deg[pos] = nre_k[k] - nrs_k[k];
              if constexpr (policy_has_accept<Policy>::value) pol.on_accept(n_k[k]);
            }
          }
        }
        if constexpr (SIDE) {
#pragma unroll
          for (int k = 0; k < ADV_ITEMS; ++k) {
            const bool aside = code_k[k] == 2;
            const unsigned long long ms = dev::ballot(aside);
            if (ms) {
              int at = 0;
              if (lane == 0) at = atomicAdd(&sm.side_cnt, __popcll(ms));
              at = dev::wave_bcast0(at);
              if (aside) sm.side[at + dev::mask_rank(ms)] = n_k[k];
            }
          }
        }
        __syncthreads();
        if constexpr (SIDE) {
          const int sc = sm.side_cnt;
          if (sc >= ADV_BLOCK) {  // flush the side pile before it could overflow on the next chunk
            if (tid == 0) sm.side_base = pol.side_reserve(c, sc);
            __syncthreads();
            const int sb = sm.side_base;
            if (sb >= 0) side_flush(pol, sm.side, sb, sc);
            __syncthreads();
            if (tid == 0) sm.side_cnt = 0;
            __syncthreads();
          }
        }
        int cnt = ad.cnt;
        if (cnt >= TILE) {  // the last k * TILE entries leave, the first cnt % TILE stay
          const int k = cnt / TILE;
          mid_flush(qout, aux_out, cnt_out, sm, cnt - k * TILE, k * TILE);
          cnt -= k * TILE;
        }
        if (tid == 0) ad.cnt = cnt;
        __syncthreads();
      }
      __syncthreads();  // seg / start / src are rewritten by the next block of slots
      }
      __syncthreads();  // tcount is rewritten by the next batch of tile counts
    }
    {
      const int rem = ad.cnt;
      if (rem > 0) mid_flush(qout, aux_out, cnt_out, sm, 0, rem);
    }
    if constexpr (SIDE) {
      const int sc = sm.side_cnt;
      if (sc > 0) {
        if (tid == 0) sm.side_base = pol.side_reserve(c, sc);
        __syncthreads();
        const int sb = sm.side_base;
        if (sb >= 0) side_flush(pol, sm.side, sb, sc);
        __syncthreads();
      }
    }
    if (!mid_barrier(c, G, epoch, sm)) {
      if (tid == 0) {
        c->mid_err = 1;
        c->done = 1;
        a.mailbox[10] = 1;
        __threadfence_system();
        a.mailbox[0] = 1;
      }
      return;
    }
    if (tid == 0) sm.n_next = __hip_atomic_load(cnt_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    n_in = sm.n_next;
    first = false;
    ++level;
    __syncthreads();
    if (n_in == 0 || n_in > MID_EXIT_V) break;
  }
  // ---- leaving: this workgroup's share of the counters, then either the end of the search or a hand-back
  if (tid == 0 && (my_edges | my_vertices)) {
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(&c->edges_visited), (unsigned long long)my_edges,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(&c->vertices_visited), (unsigned long long)my_vertices,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (n_in == 0 && !policy_drained_is_done<Policy>::value) {
    // the bucket is drained, not the search: hand an EMPTY frontier of `level` back to the head kernel
    if (w == 0 && tid == 0) {
      // ctrl.mode stays 3 until the launch is over: workgroups of this grid that START late still read it, and
      // must take the "registration closed, leave" exit -- on mode 0 they would run advance_block on a control
      // block that is being rewritten under them (the cause of a sporadic memory fault while this was `mode = 0`).
      // The next head kernel sets the mode of its level afresh.
      c->n_tiles[level & 1] = 0;
      c->n_tiles[(level & 1) ^ 1] = 0;
      c->level = level - 1;
    }
    return;
  }
  if (n_in == 0) {
    // every workgroup's counters must have landed before the leader publishes them
    if (!mid_barrier(c, G, epoch, sm)) {
      if (tid == 0) { c->mid_err = 1; c->done = 1; a.mailbox[10] = 1; __threadfence_system(); a.mailbox[0] = 1; }
      return;
    }
    if (w == 0 && tid == 0) {
      c->done = 1;
      c->level = level;
      // the counters were updated in the home L2: read them past the L1
      long long* mb64 = reinterpret_cast<long long*>(a.mailbox + 4);
      mb64[0] = (long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&c->edges_visited), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
      mb64[1] = (long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&c->vertices_visited), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
      mb64[2] = (long long)wall_clock64() - c->t_start;
      a.mailbox[1] = level;
      a.mailbox[11] = c->bin_want;  // (as publish_done: the host's hint for the next forward search on the graph)
      a.mailbox[12] = 1;            // the search ended in a LEVEL kernel (publish_done, the head kernels: 0)
      __threadfence_system();
      a.mailbox[0] = 1;
    }
    return;
  }
  // ---- hand the flat queue of `level` (n_in entries, dense) back as tiles with their metadata
  {
    const int p = level & 1;
    int32_t* q = a.frontier[p];
    const int tiles = (n_in + TILE - 1) / TILE;
    for (int t = w; t < tiles; t += G) {
      const int slot = t * TILE + tid;
      int v = -1;
      if (slot < n_in) v = __hip_atomic_load(&q[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else q[slot] = -1;
      int deg = 0;
      if (v >= 0) deg = a.ro[v + 1] - a.ro[v];
      int tot;
      (void)dev::block_exclusive_sum<ADV_BLOCK>(deg, ad.wave, &tot);
      if (tid == 0) {
        a.tile_sums[t] = tot;
        a.tile_chunks[t] = (tot + CHUNK - 1) / CHUNK;
        a.tile_count[t] = min(TILE, n_in - t * TILE);
      }
    }
    if (w == 0 && tid == 0) {
      c->n_tiles[p] = tiles;
      c->n_tiles[p ^ 1] = 0;
      c->level = level - 1;  // the next head plans `level` (and sets ctrl.mode: it must stay 3 while late workgroups start)
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// SECOND VERSION.  Same role, same entry / exit conditions, three dependent round trips fewer per level:
//   * no reservation atomic: workgroup w appends to its PRIVATE region [w * MID_SEG, (w + 1) * MID_SEG) of the next
//     queue (what does not fit -- never on the graphs this body is for -- goes to a shared overflow area behind the
//     regions, reserved with an atomic as before);
//   * the barrier IS the count exchange: a workgroup publishes {epoch, entries it appended} in ONE 64-bit store to
//     its flag word, wave 0 of every workgroup polls the <= 32 flag words (one load instruction) until all carry the
//     epoch, and the counts it has just read give the layout of the next level's input -- no arrival atomic, no
//     counter to read afterwards.  Two sets of flag words (epoch parity): a fast workgroup can only be one barrier
//     ahead of a slow one;
//   * the row offsets of the accepted vertices are loaded for ALL neighbours of the chunk, in the shadow of the claim
//     atomics, instead of after them; policies whose labels are final when claimed (BFS, SSSP with equal weights)
//     pass the label along with the entry, so the next level does not load it.
// Entry g of the level's input is found through the 32 region counts (5-step search in LDS).
template <class Policy>
__device__ __forceinline__ bool mid_exchange(const pipe_args& a, int G, int w, int& epoch, unsigned word,
                                             const int* ovf_cnt, mid_smem<Policy>& sm) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores have been acknowledged (they sit in the home L2)
  __syncthreads();
  ++epoch;
  unsigned long long* fl = a.mid_flags + (size_t)(epoch & 1) * MID_WGS;
  if (threadIdx.x == 0)
    __hip_atomic_store(&fl[w], ((unsigned long long)(unsigned)epoch << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (threadIdx.x < 64) {
    const int lane = dev::lane_id();
    unsigned long long v = 0ull;
    int spins = 0, ok = 1;
    for (;;) {
      if (lane < G) v = __hip_atomic_load(&fl[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // past the L1
      const bool late = lane < G && (unsigned)(v >> 32) != (unsigned)epoch;
      if (dev::ballot(late) == 0ull) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > MID_SPIN_LIMIT) { ok = 0; break; }
    }
    // word: entries appended (15 bits: <= MID_SEG) | out-degree sum of those entries >> 7, saturating (16 bits) | overflow flag
    const int cnt = lane < G ? (int)(v & 0x7fffull) : 0;
    const int e128 = lane < G ? (int)((v >> 15) & 0xffffull) : 0;
    const bool ovf = lane < G && ((v >> 31) & 1ull) != 0ull;
    const int inc = dev::wave_inclusive_sum(cnt);
    if (lane < MID_WGS) sm.seg_pre[lane] = inc - cnt;  // (lanes >= G: the total)
    if (lane == MID_WGS - 1) sm.seg_pre[MID_WGS] = inc;
    const bool any_ovf = dev::ballot(ovf) != 0ull;
    const int e_next = dev::wave_sum(e128);
    if (lane == 0) {
      sm.ok = ok;
      sm.next_edges = e_next;
      sm.n_ovf = any_ovf ? __hip_atomic_load(ovf_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
  }
  __syncthreads();
  return sm.ok != 0;
}

template <class Policy>
GRX_MID_FN void mid_levels_body2(const pipe_args& a, ctrl_t* c, Policy& pol, mid_smem<Policy>& sm,
                                                 const level_head& h, uint32_t xcc_mask) {
  constexpr bool SIDE = policy_has_side<Policy>::value;
  constexpr bool CARRY = policy_carries_state<Policy>::value;
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  // ---- who takes part: as in the first version
  const unsigned my_xcc = (unsigned)__builtin_amdgcn_s_getreg(0x1814) & 15u;
  if (my_xcc != (unsigned)__builtin_ctz(xcc_mask ? xcc_mask : 1u)) return;
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(&c->mid_reg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    sm.rank = (old >> 31) ? -1 : (int)(old & 0xffffu);
  }
  __syncthreads();
  const int w = sm.rank;
  if (w < 0) return;
  if (tid == 0) {
    int g = 0;
    if (w == 0) {
      const int expect = min(MID_WGS, max(1, (int)gridDim.x / max(1, __popc(xcc_mask))));
      for (int i = 0; i < 64; ++i) {
        if ((int)(__hip_atomic_load(&c->mid_reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu) >= expect) break;
        __builtin_amdgcn_s_sleep(8);
      }
      const unsigned old = __hip_atomic_fetch_or(&c->mid_reg, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      g = min((int)(old & 0xffffu), MID_WGS);
      __hip_atomic_store(&c->mid_G, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      int spins = 0;
      while ((g = __hip_atomic_load(&c->mid_G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > MID_SPIN_LIMIT) { g = -1; break; }
      }
    }
    sm.n_next = g;
  }
  __syncthreads();
  const int G = sm.n_next;
  __syncthreads();
  auto fail_out = [&]() {
    if (tid == 0) { c->mid_err = 1; c->done = 1; a.mailbox[10] = 1; __threadfence_system(); a.mailbox[0] = 1; }
  };
  if (G < 0) { fail_out(); return; }
  if (w >= G) return;
  if constexpr (policy_has_l2_local<Policy>::value) pol.l2_local = 1;
  advance_smem<Policy>& ad = sm.adv;
  int level = h.level;
  int epoch = 0;
  bool first = true;                                   // the entering frontier is a tile queue (slots may be -1)
  int n_in = ((level & 1) ? h.nt1 : h.nt0) * TILE;     // slots to look at
  int n_priv = 0;                                      // of which in the private regions (levels after the first)
  int my_pub = 0;                                      // entries of MY region in the current input queue
  unsigned t_edges = 0u;  // out-edges / vertices THIS THREAD expanded on the levels after the entering one (summed when the launch ends)
  int t_vertices = 0;
  // tuning aid (GRX_MID_DEBUG=1): wall-clock ticks the leader spends per phase, summed over the levels of the launch ->
  // ctrl.spare[0..3] {input staged | column indices + claims + compaction | flush | exchange}, spare[4] += levels
#ifdef GRX_MID_TIMERS
  const bool dbg = (a.mid_version & 0x100) != 0 && w == 0 && tid == 0;
#else
  constexpr bool dbg = false;
#endif
  long long dbg_ph[4] = {0, 0, 0, 0}, dbg_t = dbg ? (long long)wall_clock64() : 0ll;
  // -DGRX_MID_TIMERS=2: the sub-phases of a level, every one behind a FULL wait of every thread (so the loads a phase issues are
  // no longer in the shadow of the next phase's: the sum is longer than a level of the product build -- what it shows is the
  // latency of each dependent step) -> ctrl.dbg_fine[0..7]
#if defined(GRX_MID_TIMERS) && GRX_MID_TIMERS == 2
  long long dbg_f[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_ft = 0ll;
  auto fine_mark = [&](int i) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (dbg) {
      const long long now = (long long)wall_clock64();
      if (i >= 0) dbg_f[i] += now - dbg_ft;
      dbg_ft = now;
    }
  };
#else
  auto fine_mark = [&](int) {};
#endif
  int dbg_levels = 0;
  auto dbg_mark = [&](int i) {
    if (dbg) {
      const long long now = (long long)wall_clock64();
      dbg_ph[i] += now - dbg_t;
      dbg_t = now;
    }
  };
  if (tid == 0) { ad.cnt = 0; sm.side_cnt = 0; sm.out_edges = 0; sm.ovf = 0; }
  __syncthreads();
  for (;;) {
    const int p = level & 1;
    const int32_t* qin = a.frontier[p];
    int32_t* qout = a.frontier[p ^ 1];
    const int4* aux_in = reinterpret_cast<const int4*>(a.mid_aux2) + (size_t)p * MID_AUX2_CAP;
    int4* aux_out = reinterpret_cast<int4*>(a.mid_aux2) + (size_t)(p ^ 1) * MID_AUX2_CAP;
    int* ovf_cnt = &c->mid_cnt[(level + 1) % 3];
    if (w == 0 && tid == 0) __hip_atomic_store(&c->mid_cnt[(level + 2) % 3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    pol.set_level(level);
    if (epoch == 0) fine_mark(-1);
    // (the level's LDS counters were zeroed behind the previous level's exchange, by thread 0 -- every use of them below sits
    // behind at least one barrier of this level)
    // Accepted neighbours leave for the next queue AT ONCE, from registers (round 6; they used to be staged in LDS and copied out
    // by a flush behind one more barrier): a wave reserves positions for the accepted atoms of all its 8 item groups with ONE LDS
    // atomic on the level's running count, position i of the level lives at w * MID_SEG + i, whatever does not fit the region goes
    // to the shared overflow area entry by entry (never on the graphs this body is for).
    const int out_base = w * MID_SEG;
    const int out_cap = a.mid_seg_cap;
    int dsum = 0;  // out-degrees of the entries this thread appended
    auto append = [&](int pos, int v, int rs, int dg, int stbits) {
      int addr = out_base + pos;
      if (pos >= out_cap) {
        addr = MID_OVF_BASE + __hip_atomic_fetch_add(ovf_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sm.ovf = 1;
      }
      qout[addr] = v;
      if (addr < MID_AUX2_CAP) aux_out[addr] = make_int4(rs, dg, stbits, 0);
      dsum += dg;
    };
    const int n_blocks = (n_in + TILE - 1) / TILE;
    for (int i0 = 0; w + i0 * G < n_blocks; i0 += first ? ADV_BLOCK : 1) {
      int span = 1;
      if (first) {
        const int t = w + (i0 + tid) * G;
        sm.tcount[tid] = t < n_blocks ? a.tile_count[t] : 0;
        span = ADV_BLOCK;
        __syncthreads();
      }
      for (int j = 0; j < span; ++j) {
        const int blk = w + (i0 + j) * G;
        if (blk >= n_blocks) break;
        if (first && sm.tcount[j] == 0) continue;  // uniform
        int v = -1, rs = 0, deg = 0;
        typename Policy::src_state st{};
        if (first) {
          if (tid < sm.tcount[j]) v = qin[blk * TILE + tid];  // tiles are front-packed
          if (v >= 0) {
            rs = a.ro[v];
            deg = a.ro[v + 1] - rs;
            st = pol.load_source(v);
          }
        } else {
          const int g = blk * TILE + tid;
          if (g < n_in) {
            int addr;
            if (g < n_priv) {
              int s = 0;
#pragma unroll
              for (int step = MID_WGS / 2; step >= 1; step >>= 1)
                if (sm.seg_pre[s + step] <= g) s += step;
              addr = s * MID_SEG + (g - sm.seg_pre[s]);
            } else {
              addr = MID_OVF_BASE + (g - n_priv);
            }
            v = __hip_atomic_load(&qin[addr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (addr < MID_AUX2_CAP) {
              // 16 bytes written by the producer before its flag: two 8-byte loads past the L1
              const long long lo8 = __hip_atomic_load(reinterpret_cast<const long long*>(&aux_in[addr]), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
              rs = (int)(lo8 & 0xffffffffll);
              deg = (int)(lo8 >> 32);
              bool have = false;
              if constexpr (CARRY) {
                if (pol.carry_state()) {
                  const int sb = __hip_atomic_load(reinterpret_cast<const int*>(&aux_in[addr]) + 2, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                  st = pol.state_from_bits(sb);
                  have = true;
                }
              }
              if (!have) st = pol.load_source(v);
            } else {
              rs = a.ro[v];
              deg = a.ro[v + 1] - rs;
              st = pol.load_source(v);
            }
          }
        }
        fine_mark(0);  // queue entry, {row start, degree}, label: loaded
        if (!first) {  // (the head accounted for the entering level)
          t_edges += deg;
          t_vertices += v >= 0 ? 1 : 0;
        }
        // A block whose vertices all have at most MID_TM_ITEMS out-edges -- every block of a road network -- is expanded THREAD-MAPPED
        // (round 6): a lane walks the row of its own vertex, so there is no degree scan, nothing staged in LDS and no owner search
        // (8 dependent LDS reads per atom: 1.14 us of a 7.25 us level on the road stand-in, the scan 0.58 -- profiles/r6_c15_*).
        // The balanced walk below is for blocks that hold a longer row.
        const bool tm = __syncthreads_or(deg > MID_TM_ITEMS) == 0;
        int tot = 0;
        if (!tm) {
          const int ex = dev::block_exclusive_sum<ADV_BLOCK>(deg, ad.wave, &tot);
          ad.seg[tid] = ex;
          ad.start[tid] = rs;
          ad.src[tid] = v;
          ad.state[tid] = st;
          if (tid == 0) ad.seg[TILE] = tot;
          __syncthreads();
        }
        dbg_mark(0);
        fine_mark(1);  // block scan, LDS staged
        // One chunk: every phase issues the operations of all IT items of a lane before any result is used.  Two builds of it
        // (round 6): 8 items over the staged block (2048 atoms, balanced as in the level kernels) and the thread-mapped one.
        // (Measured on the way: guarding item groups that hold no atom with a uniform `k < kmax` inside ONE build turned every
        // guarded load into a wait at the join -- 83.5 -> 99.6 ms on the weighted road stand-in, profiles/r6_c11_*; a second
        // balanced build of 3 items for blocks of <= 768 out-edges: 83.8 -> 81.4 ms weighted, nothing on unit weights, r6_c12_*.)
        auto do_chunk = [&](auto items_c, auto tm_c, const int a0, const int a_end) {
          constexpr int IT = decltype(items_c)::value;
          constexpr bool TM = decltype(tm_c)::value;  // thread-mapped: item k of a lane is out-edge k of its own vertex
          int e_k[IT], slot_k[IT], n_k[IT], cand_k[IT];
#pragma unroll
          for (int k = 0; k < IT; ++k) {
            e_k[k] = -1;
            slot_k[k] = 0;
            n_k[k] = 0;
            cand_k[k] = 0;
            if constexpr (TM) {
              if (k < deg) e_k[k] = rs + k;
            } else {
              const int atom = a0 + k * ADV_BLOCK + tid;
              int lo = 0;
              if (atom < a_end) {
#pragma unroll
                for (int step = TILE / 2; step >= 1; step >>= 1)
                  if (ad.seg[lo + step] <= atom) lo += step;
                e_k[k] = ad.start[lo] + (atom - ad.seg[lo]);
              }
              slot_k[k] = lo;
            }
          }
          fine_mark(2);  // owner of every atom found (LDS search)
#pragma unroll
          for (int k = 0; k < IT; ++k)
            n_k[k] = a.ci[e_k[k] >= 0 ? e_k[k] : 0];
          bool pre_k[IT];
#pragma unroll
          for (int k = 0; k < IT; ++k) {
            pre_k[k] = false;
            {
              const bool ok = e_k[k] >= 0;
              bool pass;
              typename Policy::src_state st_k = st;
              if constexpr (!TM) st_k = ad.state[slot_k[k]];
              if constexpr (policy_has_prepare<Policy>::value)
                pass = pol.prepare(st_k, n_k[k], ok ? e_k[k] : 0, cand_k[k]);
              else
                pass = pol.precheck(st_k, n_k[k], ok ? e_k[k] : 0, cand_k[k]);
              pre_k[k] = pass & ok;
            }
          }
          fine_mark(3);  // column indices (+ weights / the read-only probe of a policy without `prepare`)
          int r1_k[IT], r2_k[IT];
#pragma unroll
          for (int k = 0; k < IT; ++k) {
            r1_k[k] = 0;
            r2_k[k] = 0;
            if (pre_k[k]) r1_k[k] = pol.claim(n_k[k], cand_k[k]);
          }
          // row offsets of every neighbour (n_k is a valid vertex even where there is no edge), issued behind the
          // claims: they are on their way while the claims are, and only the accepted ones are used
          int nrs_k[IT], nre_k[IT];
#pragma unroll
          for (int k = 0; k < IT; ++k) {
            nrs_k[k] = 0;
            nre_k[k] = 0;
            {
              nrs_k[k] = a.ro[n_k[k]];
              nre_k[k] = a.ro[n_k[k] + 1];
            }
          }
          fine_mark(4);  // first claim + row offsets of the neighbours
          if constexpr (policy_two_claims<Policy>::value) {
#pragma unroll
            for (int k = 0; k < IT; ++k) {
              const bool need = pre_k[k] & pol.need2(r1_k[k], cand_k[k]);
              if (need) r2_k[k] = pol.claim2(n_k[k]);
            }
          }
          fine_mark(5);  // second claim
          int code_k[IT];
#pragma unroll
          for (int k = 0; k < IT; ++k) {
            code_k[k] = 0;
            if (pre_k[k]) code_k[k] = pol.code(r1_k[k], r2_k[k], n_k[k], cand_k[k]);
          }
          {
            unsigned long long m_k[IT];
            int before_k[IT];
            int n_keep = 0;
#pragma unroll
            for (int k = 0; k < IT; ++k) {
              m_k[k] = dev::ballot(code_k[k] == 1);
              before_k[k] = n_keep;
              n_keep += __popcll(m_k[k]);
            }
            if (n_keep) {  // (uniform in the wave)
              int at = 0;
              if (lane == 0) at = atomicAdd(&ad.cnt, n_keep);
              at = dev::wave_bcast0(at);
#pragma unroll
              for (int k = 0; k < IT; ++k) {
                if (code_k[k] == 1) {
                  append(at + before_k[k] + dev::mask_rank(m_k[k]), n_k[k], nrs_k[k], nre_k[k] - nrs_k[k], CARRY ? cand_k[k] : 0);
                  if constexpr (policy_has_accept<Policy>::value) pol.on_accept(n_k[k]);
                }
              }
            }
          }
          if constexpr (SIDE) {
            unsigned long long m_k[IT];
            int before_k[IT];
            int n_side = 0;
#pragma unroll
            for (int k = 0; k < IT; ++k) {
              m_k[k] = dev::ballot(code_k[k] == 2);
              before_k[k] = n_side;
              n_side += __popcll(m_k[k]);
            }
            if (n_side) {
              int at = 0;
              if (lane == 0) at = atomicAdd(&sm.side_cnt, n_side);
              at = dev::wave_bcast0(at);
#pragma unroll
              for (int k = 0; k < IT; ++k)
                if (code_k[k] == 2) sm.side[at + before_k[k] + dev::mask_rank(m_k[k])] = n_k[k];
            }
            __syncthreads();
            const int sc = sm.side_cnt;
            __syncthreads();  // (every wave has read the count before any wave adds to it again)
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
        };
        if (tm) {
          do_chunk(std::integral_constant<int, MID_TM_ITEMS>{}, std::true_type{}, 0, 0);
          dbg_mark(1);
          fine_mark(6);
        } else {
          for (int a0 = 0; a0 < tot; a0 += CHUNK) {
            do_chunk(std::integral_constant<int, ADV_ITEMS>{}, std::false_type{}, a0, min(tot, a0 + CHUNK));
            dbg_mark(1);
            fine_mark(6);  // compaction, appends (stores acknowledged), side pile
          }
        }
        __syncthreads();  // seg / start / src are rewritten by the next block of slots
      }
      __syncthreads();  // tcount is rewritten by the next batch of tile counts
    }
    if constexpr (SIDE) {
      const int sc = sm.side_cnt;  // (uniform: every change of it is followed by a barrier)
      if (sc > 0) {
        if (tid == 0) sm.side_base = pol.side_reserve(c, sc);
        __syncthreads();
        const int sb = sm.side_base;
        if (sb >= 0) side_flush(pol, sm.side, sb, sc);
        if (tid == 0) sm.side_cnt = 0;
        __syncthreads();
      }
    }
    // what this workgroup appended: entries (the first out_cap of them in its region) and their out-degree sum
    auto level_totals = [&]() {
      dsum = dev::wave_sum(dsum);
      if (lane == 0 && dsum) atomicAdd(&sm.out_edges, dsum);
      dsum = 0;
      __syncthreads();
    };
    level_totals();
    int my_out = min(ad.cnt, out_cap);            // entries appended to my region this level (uniform)
    unsigned my_ovf = sm.ovf ? 0x80000000u : 0u;
    dbg_mark(2);
    static_assert(MID_SEG <= 0x7fff, "the entry count of a region fits 15 bits of the exchange word");
    const unsigned e128 = (unsigned)min(sm.out_edges >> 7, 0xffff);  // (every flush ended with a barrier)
    if (!mid_exchange(a, G, w, epoch, (unsigned)my_out | (e128 << 15) | my_ovf, ovf_cnt, sm)) { fail_out(); return; }
    dbg_mark(3);
    fine_mark(7);  // totals + exchange
    if constexpr (policy_refills<Policy>::value) {
      static_assert(SIDE, "a refilling policy keeps a side pile");
      if (a.mid_refill_max > 0 && sm.seg_pre[MID_WGS] == 0 && sm.n_ovf == 0) {
        // The frontier has drained (every workgroup read the same 32 counts).  The leader does the bucket bookkeeping -- every
        // workgroup's pile entries and their minimum landed in the home L2 before its flag did -- and its word of one more
        // exchange says whether the next bucket is pulled out of the pile here, or the launch ends as it used to.
        unsigned dec = 0u;
        if (w == 0 && tid == 0) dec = pol.refill_decide(c, a.mid_refill_max) ? 1u : 0u;
        if (!mid_exchange(a, G, w, epoch, dec, ovf_cnt, sm)) { fail_out(); return; }
        const bool go = sm.seg_pre[1] != 0;  // the leader's word
        __syncthreads();
        if (go) {
          const int n_pile = pol.refill_rebind(c);
          if (tid == 0) { ad.cnt = 0; sm.side_cnt = 0; sm.out_edges = 0; sm.ovf = 0; }
          __syncthreads();
          unsigned kmin = 0xffffffffu;
          for (int b = w; b * TILE < n_pile; b += G) {
            const int i = b * TILE + tid;
            int v = 0, cls = 0, rs = 0, dg = 0;
            unsigned key = 0u;
            if (i < n_pile) {
              v = pol.refill_entry(i);
              cls = pol.refill_class(pol.load_source(v), key);
              if (cls == 2) kmin = min(kmin, key);
            }
            if (cls == 1) {
              rs = a.ro[v];
              dg = a.ro[v + 1] - rs;
            }
            const unsigned long long mn = dev::ballot(cls == 1);
            if (mn) {
              int at = 0;
              if (lane == 0) at = atomicAdd(&ad.cnt, __popcll(mn));
              at = dev::wave_bcast0(at);
              if (cls == 1) append(at + dev::mask_rank(mn), v, rs, dg, (int)key);  // (key: the label's bits, for policies that carry it)
            }
            const unsigned long long mk = dev::ballot(cls == 2);
            if (mk) {
              int at = 0;
              if (lane == 0) at = atomicAdd(&sm.side_cnt, __popcll(mk));
              at = dev::wave_bcast0(at);
              if (cls == 2) sm.side[at + dev::mask_rank(mk)] = v;
            }
            __syncthreads();
            const int sc = sm.side_cnt;
            __syncthreads();
            if (sc >= ADV_BLOCK) {  // entries that stay: to the pile that receives appends from now on
              if (tid == 0) sm.side_base = pol.side_reserve(c, sc);
              __syncthreads();
              const int sb = sm.side_base;
              if (sb >= 0)
                for (int k = tid; k < sc; k += ADV_BLOCK) pol.refill_keep_store(sb + k, sm.side[k]);
              __syncthreads();
              if (tid == 0) sm.side_cnt = 0;
              __syncthreads();
            }
          }
          {
            const int sc = sm.side_cnt;
            if (sc > 0) {
              if (tid == 0) sm.side_base = pol.side_reserve(c, sc);
              __syncthreads();
              const int sb = sm.side_base;
              if (sb >= 0)
                for (int k = tid; k < sc; k += ADV_BLOCK) pol.refill_keep_store(sb + k, sm.side[k]);
              if (tid == 0) sm.side_cnt = 0;
              __syncthreads();
            }
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o, 64));
          if (lane == 0 && kmin != 0xffffffffu) pol.side_commit(kmin);
          level_totals();
          my_out = min(ad.cnt, out_cap);
          my_ovf = sm.ovf ? 0x80000000u : 0u;
          const unsigned e128r = (unsigned)min(sm.out_edges >> 7, 0xffff);
          if (!mid_exchange(a, G, w, epoch, (unsigned)my_out | (e128r << 15) | my_ovf, ovf_cnt, sm)) { fail_out(); return; }
        }
      }
    }
    ++dbg_levels;
    n_priv = sm.seg_pre[MID_WGS];
    n_in = n_priv + sm.n_ovf;
    my_pub = my_out;
    first = false;
    ++level;
    if (tid == 0) { ad.cnt = 0; sm.out_edges = 0; sm.ovf = 0; }  // (every thread read them in front of the exchange's barriers)
    // A level that has outgrown this body goes back to the regular kernels: too many vertices, or -- scale-free graphs: a
    // few thousand vertices a level before the hubs -- too many out-edges for the 32 workgroups of one XCD (round 4: three of
    // 16 random sources of the LJ stand-in spent 5 ms in here on levels of 10^5 vertices / 10^6 edges).
    // ... or (round 6) a next level of HUBS: the blocks of 256 entries are dealt out by entry, so a level that averages more than
    // mid_hub_deg out-edges per vertex gives single workgroups several chunks to walk one after the other (0.35 ms for one such
    // level two hops from a low-degree source of a scale-free graph, profiles/r5_c37_*); the regular level kernel spreads them
    // over the grid.  (next_edges sums per-workgroup totals >> 7: a lower bound, exact enough for a rule of thumb.)
    const long long e_next = (long long)sm.next_edges << 7;
    if (n_in == 0 || n_in > a.mid_exit_v || sm.next_edges > (a.mid_exit_e >> 7) ||
        (a.mid_hub_deg > 0 && e_next > 4096 && e_next > (long long)a.mid_hub_deg * n_in)) break;
  }
  // ---- leaving
  if (dbg) {
    for (int i = 0; i < 4; ++i) atomicAdd(&c->spare[i], (int)dbg_ph[i]);
    atomicAdd(&c->spare[4], dbg_levels);
#if defined(GRX_MID_TIMERS) && GRX_MID_TIMERS == 2
    for (int i = 0; i < 8; ++i) atomicAdd(&c->dbg_fine[i], (int)dbg_f[i]);
#endif
  }
  {
    const long long w_edges = dev::wave_sum_u32_wide(t_edges);
    const int w_vertices = dev::wave_sum(t_vertices);
    if (lane == 0 && (w_edges | w_vertices)) {
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(&c->edges_visited), (unsigned long long)w_edges,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(&c->vertices_visited), (unsigned long long)w_vertices,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (n_in == 0 && !policy_drained_is_done<Policy>::value) {
    // the bucket is drained, not the search (ctrl.mode stays 3: see the first version)
    if (w == 0 && tid == 0) {
      c->n_tiles[level & 1] = 0;
      c->n_tiles[(level & 1) ^ 1] = 0;
      c->level = level - 1;
    }
    return;
  }
  if (n_in == 0) {
    // every workgroup's counters must have landed before the leader publishes them
    if (!mid_exchange(a, G, w, epoch, 0u, &c->mid_cnt[0], sm)) { fail_out(); return; }
    if (w == 0 && tid == 0) {
      c->done = 1;
      c->level = level;
      long long* mb64 = reinterpret_cast<long long*>(a.mailbox + 4);
      mb64[0] = (long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&c->edges_visited), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
      mb64[1] = (long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&c->vertices_visited), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
      mb64[2] = (long long)wall_clock64() - c->t_start;
      a.mailbox[1] = level;
      a.mailbox[11] = c->bin_want;  // (as publish_done: the host's hint for the next forward search on the graph)
      a.mailbox[12] = 1;            // the search ended in a LEVEL kernel (publish_done, the head kernels: 0)
      __threadfence_system();
      a.mailbox[0] = 1;
    }
    return;
  }
  // ---- hand the queue of `level` back as tiles: the regions ARE tiles already (MID_SEG_TILES per region, the unused
  // ones empty), the overflow area follows them; only the metadata and the padding of the last tile are missing
  {
    const int p = level & 1;
    int32_t* q = a.frontier[p];
    const int n_ovf = n_in - n_priv;
    auto one_tile = [&](int t, int n_t) {  // n_t > 0 valid entries at the front of tile t.  Block-wide.
      const int slot = t * TILE + tid;
      int v = -1;
      if (tid < n_t) v = __hip_atomic_load(&q[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else q[slot] = -1;
      int deg = 0;
      if (v >= 0) deg = a.ro[v + 1] - a.ro[v];
      int tot;
      (void)dev::block_exclusive_sum<ADV_BLOCK>(deg, ad.wave, &tot);
      if (tid == 0) {
        a.tile_sums[t] = tot;
        a.tile_chunks[t] = (tot + CHUNK - 1) / CHUNK;
        a.tile_count[t] = n_t;
      }
    };
    const int my_tiles = (my_pub + TILE - 1) / TILE;
    for (int j = 0; j < my_tiles; ++j) one_tile(w * MID_SEG_TILES + j, min(TILE, my_pub - j * TILE));
    // empty tiles: the rest of my region, and the regions nobody owns
    for (int r = w; r < MID_WGS; r += G) {
      const int j0 = r == w ? my_tiles : 0;
      if (tid >= j0 && tid < MID_SEG_TILES) {
        const int t = r * MID_SEG_TILES + tid;
        a.tile_sums[t] = 0;
        a.tile_chunks[t] = 0;
        a.tile_count[t] = 0;
      }
    }
    const int ovf_tiles = (n_ovf + TILE - 1) / TILE;
    for (int i = w; i < ovf_tiles; i += G) one_tile(MID_WGS * MID_SEG_TILES + i, min(TILE, n_ovf - i * TILE));
    if (w == 0 && tid == 0) {
      c->n_tiles[p] = MID_WGS * MID_SEG_TILES + ovf_tiles;
      c->n_tiles[p ^ 1] = 0;
      c->level = level - 1;  // the next head plans `level`
    }
  }
}

// Which body a build carries.  ONE by default (the second version): the launch-bound searches turned out to be sensitive to
// the sheer size of the level kernel's code -- with both bodies inlined (58 KB) a near-far iteration of the road stand-in
// took 42.6 us against 31.5 us with one body (45 KB), although the extra code is never executed in that run
// (tools/history/ab_mid.py, GRX_MID=0, libraries built from the same sources).  -DGRX_MID_BOTH builds both (then
// GRX_MID_VERSION=1|2 selects at run time), -DGRX_MID_FIRST_VERSION the first one only.
template <class Policy>
__device__ __forceinline__ void mid_levels_run(const pipe_args& a, ctrl_t* c, Policy& pol, mid_smem<Policy>& sm,
                                               const level_head& h, uint32_t xcc_mask) {
#if defined(GRX_MID_BOTH)
  if ((a.mid_version & 0xff) == 2) mid_levels_body2(a, c, pol, sm, h, xcc_mask);
  else mid_levels_body(a, c, pol, sm, h, xcc_mask);
#elif defined(GRX_MID_FIRST_VERSION)
  mid_levels_body(a, c, pol, sm, h, xcc_mask);
#else
  mid_levels_body2(a, c, pol, sm, h, xcc_mask);
#endif
}

}  // namespace grx
