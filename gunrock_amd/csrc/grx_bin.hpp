// grx_bin.hpp -- BINNED top-down advance: the fat levels of a forward (top-down only) BFS as two
// streaming passes instead of one random probe per edge.
//
// What it replaces in the reference: the same merge-path advance + filter pair as
// grx_frontier.hpp (operators/advance/merge_path.hxx:112-362, filter/predicated.hxx:12-39),
// for the levels where that pair is bound by random label probes.
//
// Why.  The claim-per-edge advance (advance_block) reads the label of every neighbour: 31 M
// scattered 4-byte probes on the fat level of the LJ stand-in, each moving a 64-byte sector across
// the fabric (rocprofv3, round 1: 1.88 GB of L2-miss traffic for 0.41 GB of algorithmic bytes,
// L2 hit rate 35 %, 0.54 ms).  More than 90 % of those probes only learn "already visited".
// Here a fat level runs in two kernels whose global traffic is all coalesced streams:
//   1. SCATTER (bin_scatter2_block, bfs_scatter2_kernel): the frontier is expanded exactly as in advance_block (tiles,
//      2048-edge chunks, lanes on consecutive edges), but a neighbour id is not probed -- it is appended to the BIN of its
//      vertex range (<= 256 bins) as an offset inside the bin.  A workgroup sorts 4 chunks (8192 ids) by bin in LDS and
//      writes each bin's run with ONE reservation atomic per bin and batch.  Bins are runs of GRANULES (>= 1024 vertices)
//      cut once per graph; their capacities are STATIC -- the in-edges of the vertex range -- so the bins are one E-entry
//      array that can never overflow and needs no size pass.
//   2. SWEEP (bin_sweep2_block, bfs_sweep2_kernel): one workgroup per bin (or part of a fat bin) copies the bin's slice of
//      the visited bitmap into LDS, streams the candidates through it, merges the new bits with one global atomic per word
//      and emits the discoveries in ascending vertex order as tiles of the next frontier, with their chunk-map entries.
// The head kernel chooses per level (degree sum of the frontier >= bin_args::min_edges); all other levels of the run go
// through advance_block, whose discoveries keep the same visited bitmap current (bfs_policy::on_accept).
// Round 5 removed the first generations of both kernels (round-2 scatter inside the level kernel, slice claim, first sweep,
// the 512-thread sweep geometry, pair stores): HISTORY.md section 3.2 / 3.2.1 and profiles/history/r3_ab_* are their record.
#pragma once

#include "grx_bfs_kernels.hpp"

namespace grx {

constexpr int BIN_MAX = ADV_BLOCK;   // bins: one thread of a workgroup per bin
constexpr int BIN_BATCH = 4;         // chunks sorted together (one reservation atomic per bin and batch:
                                     // a single word sustains only ~90 atomics/us)
constexpr int BIN_PAD = 32;          // ints between two counters (each in its own 128-byte line)
constexpr int BIN_SHIFT_MAX = 17;    // widest vertex range per bin: 131072 vertices = 16 KB of bitmap in LDS
constexpr int BIN_GRAN_MAX = 4096;   // granules (>= 1024 vertices each: one 128-byte line of the bitmap) per graph
constexpr int BIN_GSHIFT_MIN = 10;

struct bin_args {
  int32_t* bins;              // E entries; bin b owns [off[b], off[b + 1])
  const int32_t* off;         // nb + 1 static offsets (capacity = in-edges of the bin's vertex range)
  int32_t* fill;              // entries in bin b this level at fill[b * BIN_PAD] (zeroed by the head kernel)
  int32_t* queue;             // claim work queue head of XCD x at queue[x * BIN_PAD] (zeroed by the head kernel)
  const int32_t* v0;          // nb + 1: first vertex of each bin
  const unsigned short* g2b16;  // second scatter: granule -> bin | (index of the granule inside its bin) << 8
  int32_t local_ids;          // 1: the bins hold ids RELATIVE to the first vertex of their bin (second scatter), 0: global ids
  int32_t sweep_items;        // second sweep: work items a level is cut into at most (<= its grid: one item per workgroup)
  int32_t entry16;            // 1: the bins hold 16-BIT offsets (every bin spans <= 65536 vertices; needs local_ids), 0: 32-bit entries
  int32_t gshift;             // granule of vertex n = n >> gshift
  int32_t n_gran;
  int32_t nb;                 // bins in use (<= BIN_MAX)
  uint32_t xcc_mask;          // hardware XCC ids present on this device (census at context creation)
  int32_t n_xcd;
  long long min_edges;        // a level with at least this many out-edges is binned (0: never) ...
  unsigned* visited;          // V-bit visited bitmap of the run
  int32_t visited_words;
  int32_t* dist;
  long long* debug;           // tuning aid (GRX_BIN_DEBUG=<level>): 8 words per workgroup and phase for that level, else null
  int32_t debug_level;
  int32_t allowed;            // this launch group carries the scatter / sweep kernels
  int32_t uniform;            // > 0: every bin is the aligned range [b << uniform, (b + 1) << uniform) (13 .. 16): bin = id >> uniform
  int32_t sweep_balance;      // second sweep: size the parts so that a level is cut into sweep_items items (GRX_SW2_BALANCE)
  int32_t no_level;           // ... and NO level kernel (exact schedule of a repeated search): a level that is not over plans mode 2
  int32_t thin_div, thin_min; // thin claim-per-edge levels: thin_workgroups() (grx_bfs_kernels.hpp)
  int32_t only_finish;        // the group is its head alone (where the previous search from this source ended): plan_in::only_finish
  int32_t max_degree;         // ... and whose frontier averages at most this many out-edges per vertex
  int32_t mid_v, mid_e;       // thresholds of the many-levels-per-launch body (grx_mid.hpp), 0: off (carried here for the head kernel)
  int32_t mid_tile_e;         // ... which no level enters with a tile of more than this many out-edges (plan_in::mid_tile_e; 0: no rule)
  int32_t bin_early_div;      // early levels are binned from min_edges / bin_early_div on (plan_in::bin_early_div; <= 1: one threshold)
  int32_t static_units;       // second scatter: units strided statically over the workgroups instead of drawn from per-XCD ticket
                              // queues (the fallback when a launch cannot be trusted to put a workgroup on every XCD)
  int32_t sub_shift;          // second scatter: log2 of the sub-counters per bin (2 | 1 | 0; GRX_BIN_SUB; nb << sub_shift <= 1024)
  int32_t fault_xcd;          // test aid (GRX_SC2_FAULT_XCD=k): workgroups on dense XCD index k - 1 take no units (0: off)
  // binned RELAXATION (weighted SSSP on dense graphs, grx_relax.hpp): an entry is the 16-bit offset of the target inside its
  // bin (in `bins`) and, at the same index of rval, the tentative distance fl(dist[source] + weight) as ordered bits
  float* rdist;               // the labels: read by the scatter (sources), read and written by the sweep (targets)
  const float* rw;            // edge weights
  unsigned* rval;             // E (+ padding) tentative distances, laid out like `bins`
  int32_t* rstamp;            // per-level stamp of a vertex (parts of one bin agree on who emits an improved vertex)
  // partitioned searches (round 6, part_args): the sweep hands the new bits of words outside [part_wlo, part_whi) -- vertices of
  // other ranks -- to the outgoing bitmap instead of emitting them (null: single GPU)
  unsigned* part_send;
  int32_t part_wlo, part_whi;
};

// hardware XCC id of this wave's CU -> dense index 0 .. n_xcd - 1 (s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4))
__device__ __forceinline__ int xcd_index(uint32_t xcc_mask, int n_xcd) {
  const unsigned id = (unsigned)__builtin_amdgcn_s_getreg(0x1814) & 15u;
  const int dense = __popc(xcc_mask & ((1u << id) - 1u));
  return ((xcc_mask >> id) & 1u) ? dense : (int)(id % (unsigned)n_xcd);
}

// ---------------------------------------------------------------------------------------------------------------
// SCATTER, second version (round 3): the same multisplit on workgroups of 1024 threads.
//
// What the counters said about the first version (profiles/history/r2_bench_pmc.json, class topdown_fat): VALU 15 % busy, the
// LDS array ~45 %, HBM at a quarter of its rate -- and the waves parked in s_waitcnt / s_barrier 59 % of their cycles.
// Nothing is saturated; the kernel is a chain of ~12 barrier-separated phases per batch executed by THREE workgroups
// (12 waves) per CU, because a thread carries 32 edges (id + bin/rank) through the sort and a software pipeline four
// chunks deep: 156 VGPRs, 44 KB of LDS.  Here a batch is still 4 chunks = 8192 edges (one reservation atomic per bin
// and batch: a counter word sustains ~90 atomics/us), but it is spread over 1024 threads -- quarter q of the
// workgroup stages chunk q -- so a thread carries 8 edges: <= 64 VGPRs, two workgroups = 32 waves per CU, the
// hardware maximum, to hide the same round trips.  Changes besides the geometry:
//   * one barrier less in the owner map: the carry of the running maximum across the four waves of a quarter comes
//     from three ballots per wave over the slots' row-begin positions (highest slot that begins a row before the
//     target wave's first atom), written with the degree totals -- no second exchange;
//   * the granule table yields bin AND the vertex's offset inside its bin in ONE 16-bit LDS read; the sorted entry is
//     (bin << 24 | offset), so the copy-out finds its bin without a second table lookup, and the bins hold offsets
//     relative to the bin's first vertex -- which is what the sweep claim indexes its bitmap slice with;
//   * the owner map no longer shares its LDS with the sort buffer, so the barrier at the end of a batch is gone
//     (7 per batch instead of ~12).
// Launched as its own kernel between the level kernel and the sweep (bfs_scatter2_kernel): the other bodies of the
// level kernel are written for 256-thread workgroups.
constexpr int SC2_BLOCK = 1024;
constexpr int SC2_Q = SC2_BLOCK / TILE;  // quarters = chunks per batch
// SUB-COUNTERS (round 4).  With 16-bit entries the LJ stand-in has 76 bins: the 64 lanes of a wave hit a handful of hot bins
// several times each, and a returning LDS atomic on one word serialises (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.51 in
// profiles/history/r3_bench_pmc.json, class topdown_fat).  Every bin gets SC2_SUB counters in neighbouring words (= banks); a lane
// uses counter lane % SC2_SUB.  A bin's run in the sorted buffer is the concatenation of its sub-runs, so everything behind
// the histogram -- one reservation per bin, one `delta` per bin, the copy-out -- is unchanged.  256 bins x 4 = one counter
// per thread of the 1024-thread workgroup.  bin_args::sub_shift = 0 switches it off (GRX_BIN_SUB=1, for the A/B).
constexpr int SC2_SUB = 4;
static_assert(BIN_MAX * SC2_SUB == SC2_BLOCK, "one sub-counter per thread in the offset scan");
static_assert(SC2_Q == BIN_BATCH, "a batch is still 4 chunks");

struct bin_scatter2_smem {
  alignas(16) int dlt[SC2_Q][TILE];            // per staged slot: row start - exclusive degree prefix
  alignas(16) int wtot[SC2_Q][4];              // degree sums of the four waves of a quarter (read as one int4)
  alignas(16) int cand[SC2_Q][4][4];           // [quarter][target wave][source wave]: highest slot of the source wave that begins a row before the target's first atom
  int wave[SC2_BLOCK / 64 + 1];
  int hist[BIN_MAX * SC2_SUB];                 // per bin SC2_SUB counters: a lane counts in counter (lane % SC2_SUB) of its bin
  int off[BIN_MAX * SC2_SUB];
  int delta[BIN_MAX];
  int btot;
  int tick[4];                                 // units of the pipeline stages (tick[3]: the stage entering next)
  unsigned g2d[BIN_GRAN_MAX];                  // granule -> (bin << BSHIFT) - first vertex of the bin: entry = id + g2d[granule of id]
  alignas(8) unsigned char own[SC2_Q][CHUNK];  // owner map: staged slot of every atom of the four chunks (8 bytes per thread)
  alignas(16) unsigned sorted[SC2_Q * CHUNK];  // (bin << 24 | offset inside the bin), grouped by bin
};


// VAL build (binned relaxation, grx_relax.hpp): every entry carries a 32-bit value through the sort, and the bin index is
// 10 bits wide (up to BIN_MAX * SC2_SUB = 1024 bins of <= 16384 vertices, one histogram counter each: no sub-counters)
struct bin_scatter2_val_smem : bin_scatter2_smem {
  unsigned dsrc[SC2_Q][TILE];                  // per staged slot: label of the slot's vertex
  int delta_v[BIN_MAX * SC2_SUB];
  alignas(16) unsigned sortedv[SC2_Q * CHUNK]; // the values, in the order of `sorted`
};

// E16: the bins hold 16-bit offsets (half the bytes written here and streamed by the sweep)
// UNI (round 5): the bins are the aligned 65536-vertex ranges (graph_build_bins: the width cap binds everywhere on a graph of a
// few million vertices, e.g. 74 of the LJ stand-in's 76 balanced bins were that already), so the bin of a target is id >> 16 and
// its offset id & 0xffff -- the sorted entry is the id itself: no granule-table read (a random 16-bit LDS read per edge, the most
// conflict-prone of the eight LDS operations an edge costs) and none of the six VALU instructions that packed the entry.
template <bool DBG, bool E16, bool VAL = false, class SM = bin_scatter2_smem, bool UNI = false>
__device__ __forceinline__ void bin_scatter2_block(const pipe_args& a, const bin_args& bn, SM& sm, int p,
                                                   int total_chunks, const int* chunk_tile) {
  static_assert(!VAL || E16, "values travel with 16-bit offsets");
  static_assert(!UNI || E16, "uniform bins: 16-bit offsets");
  constexpr int BBITS = VAL ? 10 : 8;    // bits of a bin index in the granule table
  constexpr int BSHIFT_C = VAL ? 14 : 24;  // ... and where it sits in a sorted entry (above the offset inside the bin)
  const int BSHIFT = UNI ? bn.uniform : BSHIFT_C;   // (UNI: a uniform run-time shift, the entry is the id)
  const unsigned umask = (1u << (UNI ? bn.uniform : 16)) - 1u;
  constexpr unsigned BMASK = (1u << BBITS) - 1u;
  // DBG (GRX_BIN_DEBUG, its own kernel build): wave 0's clock at the end of every phase, summed per workgroup
  long long dbg_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_t = 0, dbg_t0 = 0;
  int dbg_batches = 0;
  auto dbg_mark = [&](int i) {
    if constexpr (DBG) {
      const long long now = (long long)wall_clock64();
      dbg_ph[i] += now - dbg_t;
      dbg_t = now;
    }
  };
  if constexpr (DBG) dbg_t0 = dbg_t = (long long)wall_clock64();
  const int tid0 = threadIdx.x;
  int tid = tid0;
  int q = tid >> 8;          // quarter of the workgroup = chunk of the batch
  int tq = tid & (TILE - 1); // slot of the staged tile
  int lane = tid & 63;
  int wq = (tid >> 6) & 3;   // wave inside the quarter
  const int32_t* in = a.frontier[p];
  const int gshift = bn.gshift;
  if constexpr (!UNI) {
    // Round 5: the table holds, per granule, what turns an id into its sorted entry with ONE addition -- (bin << BSHIFT) minus
    // the bin's first vertex: id + that = bin << BSHIFT | offset inside the bin (offsets stay below 2^16 <= 2^BSHIFT, nothing
    // carries).  The 16-bit table of round 3 (bin | granule index inside the bin) cost two masks, two shifts and an or3 per edge
    // behind the same LDS read; the scatter is bound by its instruction count (profiles/r5_c4_*).
    for (int gi = tid; gi < bn.n_gran; gi += SC2_BLOCK) {
      const unsigned t = bn.g2b16[gi];
      sm.g2d[gi] = ((t & BMASK) << BSHIFT_C) - ((unsigned)(gi - (int)(t >> BBITS)) << gshift);
    }
  }
  const int sub_shift = bn.sub_shift;                     // 2: four sub-counters per bin, 0: one
  const int sub_mask = (1 << sub_shift) - 1;
  const int boff = (tid >> sub_shift) < bn.nb ? bn.off[tid >> sub_shift] : 0;   // static offset of the bin of counter `tid`
  const int n_units = (total_chunks + SC2_Q - 1) / SC2_Q;
  // DYNAMIC unit hand-out, one queue per XCD.  With units strided statically over the workgroups the last workgroup of
  // a fat level finished 17-36 us after the average one (timeline of round 3, call 3: busy mean 79 / 114 us, span 96 /
  // 150 us): the units are equal, the workgroups' speeds are not.  Unit u belongs to XCD u % n_xcd; ticket k of XCD x
  // is unit x + n_xcd * k, drawn with an L2-LOCAL atomic (workgroup scope: every taker of that word runs on that XCD;
  // one device-wide word would serve ~88 tickets/us for ~45 tickets/us of demand).  The head kernel zeroes the words.
  // SAFETY NET: the queues only work if every XCD of the census runs at least one workgroup of this launch
  // (CU masks, another partition mode, a tiny grid ...).  The sweep that follows checks that the bins received EXACTLY the
  // level's out-edges (bin_sweep2_block) and raises ctrl.mid_err = 2 otherwise; the host then repeats the search with
  // bin_args::static_units = 1 -- unit k of workgroup w is w + k * gridDim.x, no queue, no assumption -- and keeps that
  // mode for the context.
  const int xcd = xcd_index(bn.xcc_mask, bn.n_xcd);
  if (bn.fault_xcd == xcd + 1) return;  // (test aid: this XCD's units are lost on purpose)
  const bool stat = bn.static_units != 0;  // uniform
  const int n_xcd = stat ? (int)gridDim.x : bn.n_xcd;   // unit of ticket k: first + n_xcd * k
  const int first_unit = stat ? (int)blockIdx.x : xcd;
  int next_static = 4;  // uniform: the static "ticket" counter
  int* qhead = &bn.queue[(unsigned)(xcd * BIN_PAD)];
  if (tid0 == 0) {
    int t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = stat ? i : __hip_atomic_fetch_add(qhead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int i = 0; i < 4; ++i) sm.tick[i] = first_unit + n_xcd * t[i];
  }
  __syncthreads();
  int vzero;  // keeps the descriptor loads vector loads (a scalar load would be waited for at the next barrier)
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  const int2* map = reinterpret_cast<const int2*>(chunk_tile);
  auto S1 = [&](int u) -> int2 {
    const long long cidx = (long long)u * SC2_Q + q;
    const bool ok = u < n_units && cidx < total_chunks;
    int2 t = map[(unsigned)((ok ? (int)cidx : 0) + vzero)];
    if (!ok) t.y = -1;  // not a chunk of this level: contributes no atoms (its tile is a real one)
    return t;
  };
  // software pipeline of the front, as in the first version: descriptor -> slot -> row offsets are three dependent
  // round trips; iteration i issues the row offsets of unit i + 1, the slot of unit i + 2 and the descriptor of
  // unit i + 3 together with its own column indices
  int uA = __builtin_amdgcn_readfirstlane(sm.tick[0]), uB = __builtin_amdgcn_readfirstlane(sm.tick[1]),
      uC = __builtin_amdgcn_readfirstlane(sm.tick[2]);  // (tickets of one XCD grow: uA < uB < uC < the later ones)
  // of a descriptor {tile, chunk index inside the tile} the tile is needed only to load the slot: stages A and B
  // carry the chunk index alone
  int yA, yB;
  int2 tlC, tlD;
  int vA, vB, vC;
  int rsA, reA, rsB, reB;
  unsigned dA = 0u, dB = 0u;  // VAL: label of the slot's vertex, staged with its row offsets
  {
    const int2 tA = S1(uA), tB = S1(uB);
    tlC = S1(uC);
    yA = tA.y;
    yB = tB.y;
    vA = in[(unsigned)(tA.x * TILE + tq)];
    vB = in[(unsigned)(tB.x * TILE + tq)];
    const unsigned vv = vA >= 0 ? (unsigned)vA : 0u;
    rsA = a.ro[vv];
    reA = a.ro[vv + 1u];
    if constexpr (VAL) dA = __float_as_uint(bn.rdist[vv]);
  }
  // DRAIN THE PROLOGUE (round 5, from the ISA).  The waitcnt pass merges what is pending on the two ways into the loop
  // header: entering from here the loads above were the YOUNGEST operations in flight, so the header got
  // `s_waitcnt vmcnt(0)` -- which on the back edge means "wait for the eight copy-out stores this wave has just
  // issued" at the top of every batch.  Consuming the values here (an empty asm that reads them) puts the wait in
  // front of the loop; on the back edge everything the header reads is a batch old and needs no wait at all.
  if constexpr (VAL) asm volatile("" ::"v"(dA));
  asm volatile("" ::"v"(vA), "v"(vB), "v"(rsA), "v"(reA), "v"(tlC.x), "v"(tlC.y), "v"(yA), "v"(yB));
  while (uA < n_units) {
    // Everything derived from the thread index is RE-derived per batch from an opaque copy: left to itself the
    // compiler hoists two dozen per-thread constants (k * 256 + tq, LDS addresses, ...) out of the loop, runs out
    // of the 64 VGPRs that two workgroups per CU allow, and spills -- and a scratch reload issued behind the
    // prefetch loads waits for those loads (in-order vmcnt), which stalls the software pipeline.
    tid = tid0;
    asm volatile("" : "+v"(tid));
    q = tid >> 8;
    tq = tid & (TILE - 1);
    lane = tid & 63;
    wq = (tid >> 6) & 3;
    unsigned char* own = &sm.own[q][0];
    const int my_bin = tid >> sub_shift;  // bin of counter `tid` (counters beyond nb << sub_shift stay 0)
    // ---- phase 1: degrees, wave scan; clear the owner map and the histogram
    const bool has = yA >= 0;
    const int rs = rsA;
    const int dg = (has && vA >= 0) ? reA - rsA : 0;
    const int a0 = has ? yA * CHUNK : 0;
    // the front of the next batches: a whole batch of cover for their round trips
    {
      const unsigned vv = vB >= 0 ? (unsigned)vB : 0u;  // unconditional loads from a clamped index (vertex 0 exists)
      rsB = a.ro[vv];
      reB = a.ro[vv + 1u];
      if constexpr (VAL) dB = __float_as_uint(bn.rdist[vv]);
    }
    vC = in[(unsigned)(tlC.x * TILE + tq)];
    // drawn now, used behind the sort (unit of stage D of the NEXT batch).  Round 5, from the ISA: the compiler's atomic
    // optimizer turns a one-lane atomic on a UNIFORM address into mbcnt + atomic + readfirstlane and waits for the result on
    // the spot -- `s_waitcnt vmcnt(0)` behind the three front loads just issued: wave 0 sat out a whole memory round trip at
    // the top of every batch and the other fifteen waves waited for it at the first barrier.  An address the compiler cannot
    // prove uniform (the opaque zero below) keeps it a plain per-lane atomic whose result is waited for where it is used.
    int ticket = 0;
    if (stat) ticket = next_static++;
    else if (tid == 0) ticket = __hip_atomic_fetch_add(qhead + vzero, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int inc = dev::wave_inclusive_sum(dg);
    if (lane == 63) sm.wtot[q][wq] = inc;
    reinterpret_cast<uint2*>(own)[tq] = make_uint2(0u, 0u);
    sm.hist[tid] = 0;
    __syncthreads();
    const int uD = __builtin_amdgcn_readfirstlane(sm.tick[3]);  // written at the end of the previous batch (or at the start)
    tlD = S1(uD);
    ++dbg_batches;
    dbg_mark(0);
    // ---- phase 2: exclusive prefix inside the tile; rows mark where they begin; carries for the running maximum
    int base = 0, tot = 0;
    {
      const int4 wt = *reinterpret_cast<const int4*>(&sm.wtot[q][0]);
      const int x[4] = {wt.x, wt.y, wt.z, wt.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < wq) base += x[i];
        tot += x[i];
      }
    }
    const int ex = base + inc - dg;
    sm.dlt[q][tq] = rs - ex;
    if constexpr (VAL) sm.dsrc[q][tq] = dA;
    const int pos = ex - a0;  // where this slot's row begins inside the chunk's window
    if (dg > 0) {
      if (pos > 0) {
        if (pos < CHUNK) own[pos] = (unsigned char)tq;
      } else if (pos + dg > 0) {
        own[0] = (unsigned char)tq;  // the one row that is under way where the window begins
      }
    }
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      // highest slot of this wave whose row begins before wave w's first atom (its 512-byte share of the owner map)
      const unsigned long long m = dev::ballot(dg > 0 && pos < w * (CHUNK / 4));
      if (lane == 0) sm.cand[q][w][wq] = m ? (wq * 64 + 63 - __builtin_clzll(m)) : 0;
    }
    __syncthreads();
    dbg_mark(1);
    // ---- phase 3: running maximum over the owner map (8 bytes per thread, wave scan, carry from the ballots)
    {
      const uint2 w8 = reinterpret_cast<const uint2*>(own)[tq];
      unsigned m_k[8];
      unsigned run = 0u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned b = ((i < 4 ? w8.x : w8.y) >> ((i & 3) * 8)) & 0xffu;
        run = b > run ? b : run;
        m_k[i] = run;
      }
      const int incm = dev::wave_inclusive_max_nonneg((int)run);
      unsigned carry = (unsigned)dev::wave_shift_up1(incm);
      if (wq > 0) {
        const int4 cd = *reinterpret_cast<const int4*>(&sm.cand[q][wq][0]);
        const int c4 = max(max(cd.x, cd.y), max(cd.z, cd.w));
        carry = (unsigned)c4 > carry ? (unsigned)c4 : carry;
      }
      uint2 r8 = make_uint2(0u, 0u);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned v = m_k[i] > carry ? m_k[i] : carry;
        if (i < 4) r8.x |= v << (i * 8);
        else r8.y |= v << ((i - 4) * 8);
      }
      reinterpret_cast<uint2*>(own)[tq] = r8;
    }
    __syncthreads();
    dbg_mark(2);
    // ---- phase 4: edges of the chunk (lanes on consecutive atoms), column indices, bin + rank inside the bin.
    // Every LDS / global operation of the 8 atoms is issued UNCONDITIONALLY from a clamped index, phase by phase
    // (owner bytes -> row deltas -> column indices -> granule table -> histogram): under a per-lane condition each
    // one sits in its own basic block with its consumer and an s_waitcnt behind it -- 8 x 4 serialized round trips.
    unsigned e_k[ADV_ITEMS];  // first the neighbour id, then bin << BSHIFT | offset inside the bin
    int r_k[ADV_ITEMS];       // rank inside the bin
    unsigned v_k[VAL ? ADV_ITEMS : 1];  // VAL: first the source's label, then the tentative distance (ordered bits)
    const int n_at = has ? min(tot - a0, CHUNK) : 0;  // atoms of this chunk; atom k * TILE + tq is real iff < n_at
    {
      int ob[ADV_ITEMS];
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) ob[k] = own[k * TILE + tq];
      if constexpr (VAL) {
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) v_k[k] = sm.dsrc[q][ob[k]];
      }
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) ob[k] = sm.dlt[q][ob[k]];
      float w_k[VAL ? ADV_ITEMS : 1];
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        const int al = k * TILE + tq;
        const int e = al < n_at ? a0 + al + ob[k] : 0;  // lanes past the end read edge 0
        e_k[k] = (unsigned)a.ci[(unsigned)e];  // (unsigned: a 32-bit offset on the scalar base, no 64-bit address arithmetic)
        if constexpr (VAL) w_k[k] = bn.rw[(unsigned)e];
      }
      if constexpr (VAL) {
        // the relaxation's arithmetic, exactly (sssp.hxx:121-123): fl(label of the source + weight)
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) v_k[k] = __float_as_uint(__uint_as_float(v_k[k]) + w_k[k]);
      }
      if constexpr (DBG) {  // split the phase where the column indices have arrived
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg_mark(3);
      }
      if constexpr (UNI) {
        // the entry is the id: bin = id >> 16 (a lane past the end holds column 0's id: a valid bin, counted as 0)
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k)
          r_k[k] = atomicAdd(&sm.hist[((e_k[k] >> BSHIFT) << sub_shift) | (unsigned)(lane & sub_mask)], (k * TILE + tq) < n_at ? 1 : 0);
      } else {
        unsigned t_k[ADV_ITEMS];
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) t_k[k] = sm.g2d[e_k[k] >> gshift];
#pragma unroll
        for (int k = 0; k < ADV_ITEMS; ++k) {
          e_k[k] += t_k[k];
          r_k[k] = atomicAdd(&sm.hist[((e_k[k] >> BSHIFT_C) << sub_shift) | (unsigned)(lane & sub_mask)], (k * TILE + tq) < n_at ? 1 : 0);
        }
      }
    }
    __syncthreads();
    dbg_mark(4);
    // (Measured and removed, call 7 of round 3: touching the first and last column-index line of the NEXT batch's rows here,
    // half a batch ahead -- the fat level of short rows pays a cold line per row at the top of phase 4.  Slower on every
    // graph (LJ fat levels 136 + 168 -> 157 + 177 us): the reservation atomic below is younger than those loads, and
    // waiting for its result means waiting for them too -- vmcnt retires in order.)
    // ---- phase 5: one reservation per non-empty bin (its round trip is covered by the scan and the sort), bin offsets
    // (one counter per thread; the counters of a bin sit in neighbouring lanes: its total comes from two shuffles)
    const int cnt = sm.hist[tid];
    int gbase = 0;
    {
      int bt = cnt;
      if (sub_shift >= 1) bt += dev::lane_xor1(bt);  // (uniform)
      if (sub_shift >= 2) bt += dev::lane_xor2(bt);
      if ((tid & sub_mask) == 0 && bt > 0) gbase = atomicAdd(&bn.fill[(unsigned)(my_bin * BIN_PAD)], bt);
    }
    const int inc2 = dev::wave_inclusive_sum(cnt);
    if (lane == 63) sm.wave[tid >> 6] = inc2;
    __syncthreads();
    int ex2;
    {
      // totals of the waves before mine: lane l < 16 reads one, a wave sum adds the first (tid >> 6) of them (16 loads into
      // 16 registers per thread spilled this kernel past its 64 VGPRs)
      const int b2 = dev::wave_sum((lane < (tid >> 6)) ? sm.wave[lane & 15] : 0);
      ex2 = b2 + inc2 - cnt;
      sm.off[tid] = ex2;
      if (tid == SC2_BLOCK - 1) sm.btot = ex2 + cnt;
    }
    __syncthreads();
    dbg_mark(5);
    // ---- phase 6: group by bin in LDS (offsets read unconditionally, then the stores)
    {
      int o_k[ADV_ITEMS];
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) o_k[k] = sm.off[((e_k[k] >> BSHIFT) << sub_shift) | (unsigned)(lane & sub_mask)];
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k)
        if (k * TILE + tq < n_at) {
          sm.sorted[o_k[k] + r_k[k]] = e_k[k];
          if constexpr (VAL) sm.sortedv[o_k[k] + r_k[k]] = v_k[k];
        }
    }
    // global slot of sorted position i of this bin: delta + i
    if constexpr (VAL) {
      if ((tid & sub_mask) == 0) sm.delta_v[my_bin] = boff + gbase - ex2;
    } else {
      if ((tid & sub_mask) == 0 && my_bin < BIN_MAX) sm.delta[my_bin] = boff + gbase - ex2;
    }
    // the next batch's stage D: the ticket arrived long ago (it is older than the column-index loads), nothing else of this
    // wave is in flight here, so the wait is free -- and it must not sit behind the copy-out stores below.  Read after the
    // next batch's first barrier.
    if (tid == 0) sm.tick[3] = first_unit + n_xcd * ticket;
    __syncthreads();
    dbg_mark(6);
    // ---- phase 7: runs leave LDS as contiguous segments (no barrier behind it: the next batch touches the sort
    // buffer and `delta` only after six more barriers).  Positions past the batch's total hold stale entries: their
    // bin field is < 256 whatever they are, so the table read stays unconditional
    if constexpr (VAL) {
      // offsets and values leave at the same index of their arrays.  (A stale position past the batch's total may hold any
      // bits: its bin field is masked into the table.)
      const int btot = sm.btot;
      unsigned s_k[ADV_ITEMS], x_k[ADV_ITEMS];
      int d_k[ADV_ITEMS];
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        s_k[k] = sm.sorted[k * SC2_BLOCK + tid];
        x_k[k] = sm.sortedv[k * SC2_BLOCK + tid];
      }
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) d_k[k] = sm.delta_v[(s_k[k] >> BSHIFT) & BMASK];
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        const int i = k * SC2_BLOCK + tid;
        if (i < btot) {
          reinterpret_cast<unsigned short*>(bn.bins)[(unsigned)(d_k[k] + i)] = (unsigned short)(s_k[k] & (UNI ? umask : 0xffffu));
          bn.rval[(unsigned)(d_k[k] + i)] = x_k[k];
        }
      }
    } else {
      const int btot = sm.btot;
      unsigned s_k[ADV_ITEMS];
      int d_k[ADV_ITEMS];
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) s_k[k] = sm.sorted[k * SC2_BLOCK + tid];
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) d_k[k] = sm.delta[(s_k[k] >> BSHIFT) & 0xffu];  // (a stale position may hold any bits)
#pragma unroll
      for (int k = 0; k < ADV_ITEMS; ++k) {
        const int i = k * SC2_BLOCK + tid;
        if (i < btot) {
          if constexpr (UNI) reinterpret_cast<unsigned short*>(bn.bins)[(unsigned)(d_k[k] + i)] = (unsigned short)(s_k[k] & umask);
          else if constexpr (E16) reinterpret_cast<unsigned short*>(bn.bins)[(unsigned)(d_k[k] + i)] = (unsigned short)(s_k[k] & 0xffffu);
          else bn.bins[(unsigned)(d_k[k] + i)] = (int)(s_k[k] & 0xffffffu);
        }
      }
    }
    dbg_mark(7);
    uA = uB; uB = uC; uC = uD;
    yA = yB; yB = tlC.y; tlC = tlD;
    vA = vB; vB = vC;
    rsA = rsB; reA = reB;
    dA = dB;
  }
  if constexpr (DBG) {
    if (threadIdx.x == 0 && bn.debug) {
      long long* d = bn.debug + 8 * (size_t)blockIdx.x;
      d[0] = (long long)((unsigned)__builtin_amdgcn_s_getreg(0x1814) & 15u);
      d[1] = dbg_batches;
      d[2] = dbg_t0;
      d[3] = (long long)wall_clock64();
      d[4] = dbg_ph[0] | (dbg_ph[1] << 20) | (dbg_ph[2] << 40);
      d[5] = dbg_ph[3] | (dbg_ph[4] << 20) | (dbg_ph[5] << 40);
      d[6] = dbg_ph[6] | (dbg_ph[7] << 20);
      d[7] = -2;  // record of the second scatter
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// SWEEP claim, second version (round 3): the same three steps per work item (candidates -> LDS bitmap, one merging
// atomic per word with new bits, new bits -> ascending vertex ids -> labels + tiles + chunk map) on workgroups of NT
// = 512 threads and <= 64 VGPRs: FOUR workgroups (32 waves) per CU instead of one of 1024 threads at 128 VGPRs.
// Why: an item is a chain of ~8 dependent global round trips (bitmap slice, candidate stream, bitmap re-read, merging
// atomic, row offsets, chunk-map reservation, ...) of ~2 us each under load; the first version ran ONE such chain per
// CU (timeline of round 2: 23 us of streaming + 26 us of "expand" per item, 216 items on 256 CUs, the other
// workgroups idle) and ended every workgroup with a separate emission of its last short tile.  Here
//   * items are smaller: a level is cut into about as many parts as there are RESIDENT workgroups (bin_args::sweep_items,
//     at least SW2_PART_MIN candidates each), one item per workgroup, so the chains of four items per CU overlap.
//     (No device-wide queue: a single counter word serves ~88 atomics/us, 1024 workgroups asking at once would wait
//     12 us for it.)
//   * a workgroup's last item emits its short tile together with its full ones;
//   * the emission keeps one group of three passes in flight instead of all nine (registers), and issues the
//     chunk-map reservation as soon as the degree sums are known, behind the stores of the frontier slots.
constexpr int SW2_PART_MIN = 1 << 15;
constexpr int SW2_U = 2;          // 16-byte candidate loads per thread and round
#ifndef GRX_SW_RING
#define GRX_SW_RING 2
#endif
constexpr int SW2_RING = GRX_SW_RING;  // rounds of candidate loads in flight per thread (bin_sweep2_block)

template <int NT, int LIST_ENTRIES>
struct bin_sweep2_smem {
  static constexpr int SEG_WORDS = NT / 4;               // a thread expands one byte of a bitmap word
  static constexpr int LIST = LIST_ENTRIES;              // sized for ONE emission per item on the graphs measured
  static_assert(LIST_ENTRIES >= NT / 4 * 32 + TILE && LIST_ENTRIES % TILE == 0, "a segment's ids + a carried short tile fit");
  static constexpr int MAX_TILES = LIST / TILE + 1;
  unsigned bm[1 << (BIN_SHIFT_MAX - 5)];
  int list[LIST];
  int pre[BIN_MAX + 1];
  int fillv[BIN_MAX];
  int wave[NT / 64 + 1];
  int sum[MAX_TILES][4];   // per tile of an emission and wave of the tile: degree sums
  int ttot[64];
  int cpre[64];
  int tile_base;
  int chunk_base;
  int n_chunks;
};

// Emit list[0 .. n) as ceil(n / TILE) tiles of parity q (only the last one may be short) with their entries of the next
// level's chunk map and their share of its counters.  Block-wide call.
// label != nullptr (round 5): the labels of the emitted vertices are stored HERE, label[v] = depth, from the list -- lanes on
// consecutive list entries, i.e. on ascending vertex ids a few apart (~10 cache lines per store instruction); the expansion
// that built the list used to store them itself, one thread per BYTE of the bitmap (lanes 32 bytes of labels apart: a
// store instruction touched up to 32 lines, eight of them per segment and thread).
// with_map == false (partitioned runs, bfs_part_post_kernel behind a level that was NOT binned): tiles only -- the next head
// walks them like the tiles of any claim-per-edge level.
template <int NT, class S>
__device__ __forceinline__ void sweep2_emit(const pipe_args& a, ctrl_t* c, int q, S& sm, int n, int32_t* label = nullptr,
                                            int depth = 0, bool with_map = true) {
  static_assert(S::MAX_TILES <= 64, "one lane per tile of an emission");
  constexpr int PASSES = (S::LIST + NT - 1) / NT;
  constexpr int G = 3;  // passes whose row-offset loads travel together (6 measured equal: round 4, call 18)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // (re-derive per call what depends on the thread index: see bin_scatter2_block)
  const int lane = tid & 63;
  const int k = (n + TILE - 1) / TILE;
  if (tid == 0) sm.tile_base = atomicAdd(&c->n_tiles[q], k);  // travels together with the degree loads
  for (int j0 = 0; j0 < PASSES && j0 * NT < k * TILE; j0 += G) {
    int x[G], r0[G], r1[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int idx = (j0 + g) * NT + tid;
      x[g] = idx < n ? sm.list[idx] : -1;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const unsigned xx = x[g] >= 0 ? (unsigned)x[g] : 0u;  // unconditional loads from a clamped index
      r0[g] = a.ro[xx];
      r1[g] = a.ro[xx + 1u];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int idx = (j0 + g) * NT + tid;
      if (idx - lane < k * TILE) {  // wave-uniform: tiles are multiples of the wave size
        const int t = dev::wave_sum(x[g] >= 0 ? r1[g] - r0[g] : 0);
        if (lane == 0) sm.sum[idx >> 8][(idx >> 6) & 3] = t;
      }
    }
  }
  __syncthreads();
  if (tid < 64) {  // k <= MAX_TILES tiles: one lane each
    int tot = 0, ch = 0;
    if (lane < k) {
      tot = sm.sum[lane][0] + sm.sum[lane][1] + sm.sum[lane][2] + sm.sum[lane][3];
      ch = (tot + CHUNK - 1) / CHUNK;
    }
    const int inc = dev::wave_inclusive_sum(ch);
    long long es = (long long)tot;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) es += __shfl_xor(es, o, 64);
    sm.ttot[lane] = tot;
    sm.cpre[lane] = inc - ch;
    if (lane == 63) {
      sm.n_chunks = with_map ? inc : 0;
      sm.chunk_base = (with_map && inc > 0) ? atomicAdd(&c->map_chunks, inc) : 0;
    }
    if (lane == 0 && with_map) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&c->q_edges[q]), (unsigned long long)es);
      atomicAdd(&c->n_items[q], n);
    }
  }
  // frontier slots first (they need only the tile base, which arrived with the row offsets): the reservation of the
  // chunk-map entries is on its way meanwhile
  const int base = __hip_atomic_load(&sm.tile_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (int idx = tid; idx < k * TILE; idx += NT) {
    const int x = idx < n ? sm.list[idx] : -1;
    a.frontier[q][(unsigned)((base + (idx >> 8)) * TILE + (idx & (TILE - 1)))] = x;
    // (behind every load of the emission: nothing waits for these stores.  Inside the gather loop above they sat between the
    // row-offset loads and the wave sums, whose wait then covered them too: emission 10 -> 21 us on the 31 M-edge level)
    if (label && x >= 0) label[x] = depth;  // exactly one winner per vertex (bfs.hxx:117-119 assigns the same depth)
  }
  __syncthreads();
  for (int t = tid; t < k; t += NT) {
    const int tot = sm.ttot[t];
    a.tile_sums[base + t] = tot;
    a.tile_chunks[base + t] = (tot + CHUNK - 1) / CHUNK;
    a.tile_count[base + t] = min(TILE, n - t * TILE);
  }
  {
    int2* map = reinterpret_cast<int2*>(a.chunk_tile) + sm.chunk_base;
    const int nc = sm.n_chunks;
    for (int ci = tid; ci < nc; ci += NT) {
      int t = 0;  // largest t with cpre[t] <= ci (tiles without chunks are skipped over): 64 entries, 6 steps
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1)
        if (t + step < k && sm.cpre[t + step] <= ci) t += step;
      map[ci] = make_int2(base + t, ci - sm.cpre[t]);
    }
  }
  __syncthreads();
}

template <int NT, int LE, bool DBG, bool E16, bool SLICED = false>
__device__ __forceinline__ void bin_sweep2_block(const pipe_args& a, const bin_args& bn, ctrl_t* c, int depth,
                                                 bin_sweep2_smem<NT, LE>& sm, int p) {
  constexpr int EPL = E16 ? 8 : 4;  // entries per 16-byte load
  using S = bin_sweep2_smem<NT, LE>;
  // DBG (GRX_BIN_DEBUG, its own kernel build): thread 0's clock per step, summed over the workgroup's items
  long long dbg_t0 = 0, dbg_t = 0, dbg_ph[4] = {0, 0, 0, 0}, dbg_entries = 0;
  int dbg_items = 0;
  auto dbg_mark = [&](int i) {
    if constexpr (DBG) {
      const long long now = (long long)wall_clock64();
      dbg_ph[i] += now - dbg_t;
      dbg_t = now;
    }
  };
  if constexpr (DBG) dbg_t0 = (long long)wall_clock64();
  static_assert(TILE == 256 && NT == 4 * S::SEG_WORDS && NT >= BIN_MAX, "a thread expands one byte of a bitmap word");
  const int tid0 = threadIdx.x;
  int tid = tid0;
  const int q = p ^ 1;
  if (blockIdx.x == 0 && tid == 0) c->map_level = depth;  // the chunk map and the counters of level `depth` come from this kernel
  int fill = 0;
  if (tid < bn.nb) fill = bn.fill[(unsigned)(tid * BIN_PAD)];
  int tot_fill;
  (void)dev::block_exclusive_sum<NT>(fill, sm.wave, &tot_fill);
  // every out-edge of the level's frontier is one candidate: the bins must hold EXACTLY q_edges[p] entries.  Anything else
  // means the scatter lost or repeated work units (see bin_scatter2_block: per-XCD ticket queues) -- the search is
  // abandoned with an error code the host acts on instead of returning wrong depths.
  if (bn.local_ids && (long long)tot_fill != c->q_edges[p]) {
    if (blockIdx.x == 0 && tid == 0) {
      c->mid_err = 2;
      c->done = 1;
      a.mailbox[10] = 2;
      __threadfence_system();
      a.mailbox[0] = 1;
    }
    return;
  }
  // every bin rounds its number of parts up: total / (sweep_items - nb) per part keeps the item count within sweep_items
  const int parts = max(1, bn.sweep_items - bn.nb);
  int PART = max(SW2_PART_MIN, ((tot_fill / parts) + 4) & ~3);  // never more than sweep_items items: every bin rounds up once
  int tot_items;
  int ex0 = dev::block_exclusive_sum<NT>((fill + PART - 1) / PART, sm.wave, &tot_items);
  // BALANCE (round 5).  That bound is loose -- most bins hold less than one part and still count as an item -- so the level
  // came out as ~210 items on 256 CUs with the hot bins' parts a third larger than the mean, and a sweep is as slow as its
  // largest item.  Aim at sweep_items exactly: start from the mean, grow the part until the count fits (two or three block
  // scans, ~0.4 us each, the same in every workgroup).
  if (bn.sweep_balance && tot_items < bn.sweep_items) {
    int p2 = max(SW2_PART_MIN, ((tot_fill / bn.sweep_items) + 4) & ~3);
    for (int it = 0; it < 4 && p2 < PART; ++it) {
      int n2;
      const int e2 = dev::block_exclusive_sum<NT>((fill + p2 - 1) / p2, sm.wave, &n2);
      if (n2 <= bn.sweep_items) {
        PART = p2;
        tot_items = n2;
        ex0 = e2;
        break;
      }
      p2 = (int)(((long long)p2 * n2 / bn.sweep_items + 64) & ~3ll);  // (uniform: every thread holds the same counts)
    }
  }
  if (tid < BIN_MAX) {
    sm.pre[tid] = ex0;
    sm.fillv[tid] = fill;
  }
  if (tid == 0) sm.pre[BIN_MAX] = tot_items;
  __syncthreads();
  int n_list = 0;  // uniform: entries waiting in sm.list (their labels are stored when they are emitted)
  const int4* src4 = reinterpret_cast<const int4*>(bn.bins);
  for (int item = (int)blockIdx.x; item < tot_items; item += (int)gridDim.x) {
    tid = tid0;
    asm volatile("" : "+v"(tid));  // (per-thread constants are re-derived per item instead of living in VGPRs)
    if constexpr (DBG) dbg_t = (long long)wall_clock64();
    int b = 0;  // largest b with pre[b] <= item (bins without items are skipped over)
#pragma unroll
    for (int step = BIN_MAX / 2; step >= 1; step >>= 1)
      if (sm.pre[b + step] <= item) b += step;
    const int e0 = (item - sm.pre[b]) * PART;
    const int n_e = min(sm.fillv[b], e0 + PART) - e0;
    const int lo = bn.off[b] + e0, hi = lo + n_e;
    const int vbase = bn.v0[b];
    const int vsub = bn.local_ids ? 0 : vbase;  // second scatter: the candidates are offsets inside the bin already
    const int words = (bn.v0[b + 1] - vbase) >> 5;
    const int gw0 = vbase >> 5;
    // first candidates on their way while the bitmap slice is copied
    const int i4_first = lo / EPL, i4_last = (hi - 1) / EPL;
    // A RING of SW2_RING rounds of candidate loads in flight (round 5, from the ISA).  The round-3 loop read `cur = nx;
    // LOAD(r + 1, nx); process(cur)`: the compiler put the copy nx -> cur at the END of the body, behind `s_waitcnt vmcnt(0)`
    // -- every round waited out the full latency of the loads it had just issued (2 x 16 B per thread = 32 KB per CU in flight:
    // 17 us for the 300 KB of an item, 3.6 TB/s over the device), and the deeper variant of round 3 (profiles/
    // r3_ab_sweep_deeper_prefetch_rejected.txt) serialised the same way with more registers.  Here the buffers are indexed
    // by compile-time constants of a fully unrolled group of SW2_RING rounds -- no register copies -- so a round waits with
    // vmcnt((SW2_RING - 1) * SW2_U) for loads issued SW2_RING - 1 rounds ago.  Rounds past the end load a clamped index and
    // find every entry out of range.
    int4 buf[SW2_RING][SW2_U];
    auto LOAD = [&](int r, int4(&v)[SW2_U]) {
#pragma unroll
      for (int u = 0; u < SW2_U; ++u) {
        const int idx = i4_first + (r * SW2_U + u) * NT + tid;
        v[u] = src4[idx < i4_last ? idx : i4_last];
      }
    };
#pragma unroll
    for (int k = 0; k < SW2_RING; ++k) LOAD(k, buf[k]);
    for (int w = tid; w < words; w += NT) sm.bm[w] = (gw0 + w) < bn.visited_words ? bn.visited[gw0 + w] : ~0u;
    // (the slice's loads are consumed here: what the stream loop below finds in flight is the ring alone)
    __syncthreads();
    // A. candidates -> LDS bitmap
    auto PROCESS = [&](int r, const int4(&cur)[SW2_U]) {
#pragma unroll
      for (int u = 0; u < SW2_U; ++u) {
        const int idx = i4_first + (r * SW2_U + u) * NT + tid;
        const int g0 = idx * EPL;
        int n_e4[EPL];
        if constexpr (E16) {
          const unsigned q4[4] = {(unsigned)cur[u].x, (unsigned)cur[u].y, (unsigned)cur[u].z, (unsigned)cur[u].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            n_e4[2 * j] = (int)(q4[j] & 0xffffu);
            n_e4[2 * j + 1] = (int)(q4[j] >> 16);
          }
        } else {
          n_e4[0] = cur[u].x; n_e4[1] = cur[u].y; n_e4[2] = cur[u].z; n_e4[3] = cur[u].w;
        }
        unsigned wv[EPL];
        // (unconditional LDS reads from clamped positions first, then the rare atomics)
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          const int gi = g0 + j;
          const bool ok = idx <= i4_last && gi >= lo && gi < hi;
          const int local = ok ? n_e4[j] - vsub : 0;
          wv[j] = ok ? sm.bm[local >> 5] : ~0u;
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
          const int local = n_e4[j] - vsub;
          const unsigned bit = 1u << (local & 31);
          // plain read first: a visited hub is hit by many lanes at once, and a read broadcasts where an atomic on
          // one word serialises
          if (!(wv[j] & bit)) atomicOr(&sm.bm[local >> 5], bit);
        }
      }
    };
    // The LEAN body (round 5; 16-bit entries, every entry of every lane inside [lo, hi)).  The general body above spends ~20
    // instructions on an entry -- three range compares, two mask operations, the unpacking, an exec-masked read, a branch around
    // the atomic -- and the per-item clocks of round 4 (17 us for 154 k entries on ONE CU: 64 entries per ~17 clocks) are what
    // that many VALU / SALU instructions cost four waves per SIMD, not what the loads cost (more loads in flight changed
    // nothing).  All but the first and the last 16-byte load of an item are interior: there the word is read at the byte address
    // (x >> 3) & 0x1ffc straight from the packed pair, the bit tested with one v_bfe, and ONE branch per load skips the eight
    // atomics when every bit is already set (98 % of the candidates of the second fat level are visited).
    auto PROCESS_LEAN = [&](const int4(&cur)[SW2_U]) {
      const char* bmb = reinterpret_cast<const char*>(sm.bm);
#pragma unroll
      for (int u = 0; u < SW2_U; ++u) {
        const unsigned q4[4] = {(unsigned)cur[u].x, (unsigned)cur[u].y, (unsigned)cur[u].z, (unsigned)cur[u].w};
        unsigned wv[8], sh[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wv[2 * j] = *reinterpret_cast<const unsigned*>(bmb + ((q4[j] >> 3) & 0x1ffcu));
          wv[2 * j + 1] = *reinterpret_cast<const unsigned*>(bmb + ((q4[j] >> 19) & 0x1ffcu));
          sh[2 * j] = q4[j];            // (a shift uses the low five bits of its amount)
          sh[2 * j + 1] = q4[j] >> 16;
        }
        unsigned all = 1u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          wv[j] = (wv[j] >> (sh[j] & 31u)) & 1u;
          all &= wv[j];
        }
        if (!all) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned local = (j & 1) ? (q4[j >> 1] >> 16) : (q4[j >> 1] & 0xffffu);
            if (!wv[j]) atomicOr(&sm.bm[local >> 5], 1u << (local & 31u));
          }
        }
      }
    };
    const int rounds = (i4_last - i4_first + SW2_U * NT) / (SW2_U * NT);
    for (int r0 = 0; r0 < rounds; r0 += SW2_RING) {
#pragma unroll
      for (int k = 0; k < SW2_RING; ++k) {
        const int r = r0 + k;
        // uniform: the 16-byte loads of this round lie strictly inside the item for every thread (the first load of an item
        // may begin before `lo`, its last one end behind `hi`; rounds past the end are not interior either)
        const int f = i4_first + r * SW2_U * NT;
        if (E16 && vsub == 0 && f > i4_first && f + SW2_U * NT - 1 < i4_last) PROCESS_LEAN(buf[k]);
        else PROCESS(r, buf[k]);
        LOAD(r + SW2_RING, buf[k]);
      }
    }
    __syncthreads();
    if constexpr (DBG) { ++dbg_items; dbg_entries += n_e; }
    dbg_mark(0);
    // B. words with bits the global bitmap lacks: one atomic each; what it returns decides between the parts of a bin
    for (int w0 = 0; w0 < words; w0 += 4 * NT) {
      unsigned cand[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = w0 + j * NT + tid;
        cand[j] = 0u;
        if (w < words && gw0 + w < bn.visited_words) cand[j] = sm.bm[w] & ~bn.visited[gw0 + w];
      }
      unsigned old[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = w0 + j * NT + tid;
        old[j] = 0u;
        if (cand[j]) old[j] = atomicOr(&bn.visited[gw0 + w], cand[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = w0 + j * NT + tid;
        unsigned nw = cand[j] & ~old[j];
        if constexpr (SLICED) {
          // vertices of another rank: reported to their owner through the outgoing bitmap (the bit just set in `visited`
          // means "reported" for them), never emitted here
          if (nw != 0u && (gw0 + w < bn.part_wlo || gw0 + w >= bn.part_whi)) {
            (void)__hip_atomic_fetch_or(&bn.part_send[gw0 + w], nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nw = 0u;
          }
        }
        if (w < words) sm.bm[w] = nw;
      }
    }
    __syncthreads();
    dbg_mark(1);
    const bool last = item + (int)gridDim.x >= tot_items;  // uniform: this workgroup takes no further item
    // C. new bits -> ascending vertex ids -> labels, tiles.  The list is emitted when the next segment might not fit and
    // at the end of the item -- including its short tail when this is the workgroup's last item.
    auto emit_list = [&](bool all) {
      const int k = all ? (n_list + TILE - 1) / TILE : n_list / TILE;
      const int n_emit = all ? n_list : k * TILE;
      sweep2_emit<NT>(a, c, q, sm, n_emit, bn.dist, depth);  // (the labels of the emitted vertices are stored there)
      const int rem = n_list - n_emit;
      int keep = 0;
      if (tid < rem) keep = sm.list[n_emit + tid];
      __syncthreads();
      if (tid < rem) sm.list[tid] = keep;
      n_list = rem;
      __syncthreads();
    };
    // ONE SHOT (round 5) when the item's discoveries fit the list behind what is waiting there: a thread takes wpt
    // consecutive words, one block scan gives every thread its place, and the bits become ids in LDS only -- three barriers
    // per item.  The segment loop below (a block scan and three barriers per 8192 vertices of the bin's range, eight rounds
    // for a 65536-vertex bin whatever they hold) remains for items that overflow the list.
    bool expanded = false;
    {
      const int wpt = (words + NT - 1) / NT;  // uniform, <= 4
      const int w0 = tid * wpt;
      int cnt = 0;
      for (int j = 0; j < wpt; ++j) cnt += (w0 + j) < words ? __popc(sm.bm[w0 + j]) : 0;
      int tot;
      const int ex = dev::block_exclusive_sum<NT>(cnt, sm.wave, &tot);
      if (tot == 0) {
        expanded = true;
      } else if (n_list + tot <= S::LIST) {
        int pos = n_list + ex;
        for (int j = 0; j < wpt; ++j) {
          unsigned word = (w0 + j) < words ? sm.bm[w0 + j] : 0u;
          const int v_first = vbase + ((w0 + j) << 5);
          while (word) {
            sm.list[pos++] = v_first + __ffs(word) - 1;
            word &= word - 1u;
          }
        }
        n_list += tot;
        expanded = true;
        __syncthreads();
      }
    }
    for (int s0 = 0; s0 < words && !expanded; s0 += S::SEG_WORDS) {
      const int w = s0 + (tid >> 2);
      unsigned byte = w < words ? (sm.bm[w] >> ((tid & 3) * 8)) & 0xffu : 0u;
      int tot;
      const int ex = dev::block_exclusive_sum<NT>(__popc(byte), sm.wave, &tot);
      if (tot == 0) continue;
      if (n_list + tot > S::LIST) emit_list(false);  // n_list >= TILE here: tot <= LIST - TILE
      int pos = n_list + ex;
      const int v_first = vbase + (w << 5) + (tid & 3) * 8;
      while (byte) {
        sm.list[pos++] = v_first + __ffs(byte) - 1;
        byte &= byte - 1u;
      }
      n_list += tot;
      __syncthreads();
    }
    dbg_mark(2);
    if (last ? n_list > 0 : n_list >= TILE) emit_list(last);
    dbg_mark(3);
  }
  if constexpr (DBG) {
    if (threadIdx.x == 0 && bn.debug && blockIdx.x < 4096) {
      long long* d = bn.debug + 8 * (4096 + (size_t)blockIdx.x);
      d[0] = (long long)((unsigned)__builtin_amdgcn_s_getreg(0x1814) & 15u);
      d[1] = dbg_items;
      d[2] = dbg_t0;
      d[3] = (long long)wall_clock64();
      d[4] = dbg_entries;
      d[5] = dbg_ph[0] | (dbg_ph[1] << 32);   // stream | merge
      d[6] = dbg_ph[2] | (dbg_ph[3] << 32);   // expand | emit
      d[7] = -2;
    }
  }
}

// Per-graph static part: in-edges per GRANULE (the unit bins are cut from).  One pass over the column
// indices, once per graph.  <<<grid, 256>>>
static __global__ void bin_count_kernel(const int32_t* __restrict__ ci, int64_t E, int gshift, int n_gran, int32_t* cnt) {
  __shared__ int s_hist[BIN_GRAN_MAX];
  for (int i = threadIdx.x; i < BIN_GRAN_MAX; i += blockDim.x) s_hist[i] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
    const unsigned g = (unsigned)ci[e] >> gshift;  // an id outside [0, V) is left uncounted: the host sees the shortfall
    if (g < (unsigned)n_gran) atomicAdd(&s_hist[g], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_gran; i += blockDim.x) {
    const int v = s_hist[i];
    if (v) atomicAdd(&cnt[i], v);
  }
}

// Which hardware XCC ids exist on this device: every workgroup ORs its id into *mask.  <<<many, 64>>>
static __global__ void xcc_census_kernel(unsigned* mask) {
  if (threadIdx.x == 0) atomicOr(mask, 1u << ((unsigned)__builtin_amdgcn_s_getreg(0x1814) & 15u));
}

}  // namespace grx
