// grx_common.hpp -- host-side state shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/grx.h"

namespace grx {

// ---- error plumbing --------------------------------------------------------
void set_error(const std::string& msg);
grx_status_t fail(grx_status_t code, const std::string& msg);

#define GRX_HIP(expr)                                                              \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      return ::grx::fail(GRX_ERROR_HIP, std::string(hipGetErrorString(_e)) +       \
                                            "\t: " #expr " (" __FILE__ ":" +       \
                                            std::to_string(__LINE__) + ")");       \
    }                                                                              \
  } while (0)

// ---- device scratch arena ----------------------------------------------------
// One growable device allocation per purpose, kept across runs (the reference
// re-allocates frontiers and a 1-element counter on every run / every advance:
// framework/enactor.hxx:181-192, advance/block_mapped.hxx:244).
struct dbuf {
  void* ptr = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t need) {
    if (need <= bytes) return hipSuccess;
    if (ptr) {
      hipError_t e = hipFree(ptr);
      if (e != hipSuccess) return e;
      ptr = nullptr;
      bytes = 0;
    }
    size_t want = need + need / 8 + 256;
    hipError_t e = hipMalloc(&ptr, want);
    if (e != hipSuccess) return e;
    bytes = want;
    return hipSuccess;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(ptr); }
};

// Device-resident control block driving the level loop without host round
// trips.  All kernels read their sizes from here; the host only polls `done`.
struct ctrl_t {
  int32_t level;          // enactor iteration
  int32_t done;           // frontier empty / converged
  int32_t n_tiles[2];     // frontier tiles (256 slots each) per parity buffer
  int32_t n_items[2];     // valid frontier entries per parity buffer
  int32_t total_chunks;   // advance work items of the current level
  int32_t map_chunks;     // chunk map entries written by the producers of the NEXT frontier (sweep claim, grx_bin.hpp)
  int64_t edges_visited;
  int64_t vertices_visited;
  int32_t map_level;      // level whose chunk map / counters (map_chunks, n_items, q_edges) the producers wrote; else the head builds them
  int32_t spare[5];
  // PageRank scalars
  float pr_dsum;
  float pr_err;
  int32_t pr_iter;
  int32_t pad1;
  // direction-optimising BFS
  int32_t mode;             // direction of the CURRENT level: 0 top-down (queue), 1 bottom-up (bitmap)
  int32_t frontier_bitmap;  // 1 if the frontier entering the next decide step is a bitmap
  int32_t convert;          // this level: 0 none, 1 bitmap -> queue, 2 labels -> bitmaps
  int32_t bu_R;             // > 0: the frontier entering the next plan step sits in bu_R static tile ranges ...
  int32_t bu_T;             // ... of bu_T tile indices each (left by a bottom-up level); 0: dense tiles [0, n_tiles)
  int32_t pad2[1];
  int64_t q_edges[2];       // out-degree sum of the frontier entering the level (set by plan/decide)
  // near-far (delta-stepping) SSSP
  float nf_lo, nf_hi;       // current bucket [lo, hi)
  float nf_delta;
  uint32_t nf_min_far;      // min tentative distance in the far pile (ordered bits)
  int32_t nf_far_n[2];      // entries in the far piles
  int32_t nf_sel;           // far pile receiving appends
  int32_t nf_split;         // this iteration moves bucket [lo, hi) from far[sel ^ 1] to the frontier
  int32_t nf_overflow;      // far pile capacity exceeded (host reruns label-correcting)
  int32_t nf_phases;
  int64_t bu_open;          // bottom-up accounting: unvisited vertices examined (cumulative)
  int64_t bu_probes;        // bottom-up accounting: in-edges read (cumulative)
  int64_t g_edges_visited;  // partitioned BFS: out-edges of all expanded vertices, whole graph
  int64_t t_start;          // wall_clock64() at the seed of the search (init kernel)
  // many mid-size levels in one launch (grx_mid.hpp)
  int32_t mid_bar;          // arrivals at the barrier since the head kernel chose mode 3
  int32_t mid_cnt[3];       // entries in the flat queue of level L at [L % 3]
  int32_t mid_err;          // a barrier timed out (the host reports an error)
  uint32_t mid_reg;         // registrations of workgroups on the home XCD (bit 31: window closed)
  int32_t mid_G;            // number of workgroups taking part (published by the leader)
  int32_t bin_want;         // bit g: the head of launch group g found a level fat enough to be binned (grx_bin.hpp)
  int32_t dbg_fine[8];      // -DGRX_MID_TIMERS=2 builds: clock sums of the many-levels body's sub-phases (tools/mid_phases.py)
};

struct level_rec {
  int64_t frontier_size;
  int64_t edges;
  float advance_ms;
  float other_ms;
  int32_t bottom_up = 0;
  int64_t bu_open = 0;
  int64_t bu_probes = 0;
};

}  // namespace grx

struct grx_context {
  int32_t device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_order = nullptr;  // grx_context_order_after
  int32_t num_cus = 256;
  uint32_t xcc_mask = 1u;  // hardware XCC ids seen by a census at context creation
  int32_t n_xcd = 1;

  grx::ctrl_t* d_ctrl = nullptr;       // device control block
  grx::ctrl_t* h_ctrl = nullptr;       // pinned host mirror (written by memcpy)
  volatile int32_t* h_mailbox = nullptr;  // pinned, device-visible: [0]=done,[1]=level,[2]=n_items,[3]=group started,
                                          // [4..9] = int64 {edges visited, vertices visited, wall-clock ticks} at `done`
  double wall_clock_khz = 100000.0;       // rate of wall_clock64() on this device
  int64_t mailbox_ticks = 0;              // elapsed wall-clock ticks of the last search that returned through the mailbox
  int32_t* d_mailbox = nullptr;        // device pointer aliasing h_mailbox

  // scratch
  grx::dbuf frontier[2];   // vertex ids, tiled
  grx::dbuf tile_chunks;   // per tile: number of advance chunks
  grx::dbuf tile_sums;     // per tile: sum of degrees
  grx::dbuf tile_count;    // per tile: valid vertices
  grx::dbuf bu_part;       // per workgroup partial counters of the bottom-up kernel
  grx::dbuf far[2];        // near-far SSSP: far piles
  grx::dbuf chunk_tile;    // per chunk: int2 {owning tile, chunk index inside the tile}
  grx::dbuf bitmap[2];     // visited / scratch bitmaps
  grx::dbuf labels;        // int32 per vertex (SSSP stamps etc.)
  grx::dbuf fbuf[4];       // float per vertex (PR plast, iweights, x, ...)
  grx::dbuf misc;          // reductions etc.
  grx::dbuf mid_aux;       // grx_mid.hpp: row start / degree carried with the flat queue entries
  grx::dbuf bins;          // binned forward BFS levels (grx_bin.hpp): the candidate array of the level, E + 16 entries
  grx::dbuf bin_fill;      // ... its per-bin fill counters and per-XCD ticket words (per SEARCH state: lives with the context,
                           // so that two contexts may search one graph handle concurrently)

  grx::dbuf rbins[3];      // binned relaxation (grx_relax.hpp): 16-bit target offsets, 32-bit tentative distances (E + 16 entries each),
                           // fill counters + queue words
  grx::dbuf blk_buf[3];    // block-asynchronous searches (grx_block.hip): dist, expd in the block numbering; bmin + queue
  grx_block_stats_t block_stats{};  // of the last search (supersteps == 0: it did not take that path)
  bool sc2_static = false;  // the binned scatter draws its units statically (set for good once a search failed the coverage check)

  grx_run_stats_t stats{};
  std::vector<grx::level_rec> levels;
};

// A device allocation that lives as long as a scope (scratch of the per-graph builders: an early error return frees it).
struct dev_scratch {
  void* p = nullptr;
  dev_scratch() = default;
  dev_scratch(const dev_scratch&) = delete;
  dev_scratch& operator=(const dev_scratch&) = delete;
  ~dev_scratch() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
  void* release() { void* q = p; p = nullptr; return q; }  // the caller adopts the allocation
};
// The state word of a lazily built per-graph structure (0: not built, 1: usable, 2: not applicable, ...) while it is being
// built under grx_graph::prep_mu: an error return leaves it at 0, so the next call tries again (round 5; it used to be set to
// "not applicable" up front, and one transient failure disabled the path for the life of the handle).
struct lazy_state {
  int32_t* st;
  bool decided = false;
  explicit lazy_state(int32_t* s) : st(s) {}
  ~lazy_state() { if (!decided) *st = 0; }
  grx_status_t done(int32_t v, grx_status_t rc = GRX_SUCCESS) { *st = v; decided = true; return rc; }
};

// A lazy per-graph build that failed for want of memory (or on a HIP error of its own launches) is not a reason to fail the
// SEARCH: the caller runs without what the build would have provided -- slower, same result -- and the build is tried again by the
// next call (its state word went back to 0).  Hard errors (invalid input) are returned.  ADVICE r5.
inline bool build_failed_softly(grx_status_t st) {
  if (st != GRX_ERROR_OUT_OF_MEMORY && st != GRX_ERROR_HIP) return false;
  (void)hipGetLastError();
  return true;
}

struct grx_graph {
  grx_context_t ctx = nullptr;
  // Serialises the LAZY per-graph builds (transpose, bin tables, two-neighbour array, pull layouts, block structure, weight
  // statistics).  Every builder takes it before it looks at its state word, builds into locals and publishes under it, so
  // two contexts may make their first search on one handle at the same time: one builds, the other finds it built.
  // (recursive: graph_is_symmetric builds the transpose.)
  std::recursive_mutex prep_mu;
  int32_t V = 0, E = 0;
  const int32_t* ro = nullptr;
  const int32_t* ci = nullptr;
  const float* w = nullptr;  // may be null => 1.0
  int32_t directed = 1, weighted = 1, symmetric = 0;
  int32_t sym_checked = 0;  // 0: `symmetric` not verified yet, 1: CSR == its transpose, 2: it is not
  // lazily built transpose (CSC) for pull operators; owned
  int32_t* t_ro = nullptr;
  int32_t* t_ci = nullptr;
  float* t_w = nullptr;
  bool has_transpose = false;
  unsigned* closed0 = nullptr;  // bitmap: vertices without in-edges (direction-optimising BFS), built lazily; owned
  int32_t closed0_words = 0;
  const void* closed0_of = nullptr;  // the in-edge offsets it was built from (the transpose, or the rows a partition brought)
  int32_t* bu_heads = nullptr;     // {first, second in-neighbour} per vertex (bottom-up probes), built lazily; owned
  const void* bu_heads_of = nullptr;  // the in-edge array it was built from (CSR of a symmetric graph, or the transpose)
  int32_t* hf_ci = nullptr;        // partitioned searches: the column array with every row's hub entries first (grx_transpose.hip); owned
  int32_t hf_state = 0;            // 0: not built, 1: usable, 2: not available
  // binned top-down levels (grx_bin.hpp), built lazily; owned
  int32_t* bin_off = nullptr;   // static bin offsets (in-edges per vertex range)
  unsigned char* bin_tab8 = nullptr;  // granule -> bin, bin -> owning XCD
  int32_t bin_shift = 0, bin_ngran = 0, bin_nb = 0;
  std::atomic<uint32_t> bin_hint{0};  // launch groups in which forward searches on this graph met a fat level (ctrl_t::bin_want,
                                      // OR-ed over the searches; 0: none yet)
  // launch groups the previous search on this graph needed (0: none yet) -- [0] forward BFS, [1] direction-optimising BFS
  // (paced: the host enqueues that many groups and then waits for the end of the search, or for the stream to drain without
  // it, instead of queueing two more behind the running one), [2] weighted SSSP on a dense graph (blind batches: the first
  // batch is that many groups instead of 4, 8, 16, ... with a host round trip between them).  run_levels, grx_engine.hpp
  std::atomic<int32_t> group_hint[3] = {{0}, {0}, {0}};
  // EXACT schedule of a repeated forward search (round 5): {source + 1 (0: none), launch groups in which THAT search met a fat
  // level} of the last forward search with binned levels.  A search from the same source on the same handle visits the same
  // levels in the same groups, so its fat groups carry head + scatter + sweep (no level kernel: the head is told and plans
  // the binned body whatever it finds) and the other groups head + level kernel only -- no no-op launches at all.  Any other
  // source falls back to the OR-ed hint above with its slack.  Packed into one word so that concurrent searches read a pair.
  std::atomic<uint64_t> bin_exact[2] = {{0}, {0}};  // [1]: profiled searches (one launch group per level, level 0 included: other groups)
  std::atomic<int32_t> end_in_head[2] = {{0}, {0}};  // the last paced search (forward | direction-optimising) ended in a head kernel
  std::atomic<uint32_t> do_last_src{0};  // source + 1 of the last direction-optimising search (the group it ended in is launched as a head alone)
  std::atomic<int32_t> pr_iter_hint{0};  // iterations of the previous PageRank run on this handle: its first blind batch (grx_pr.hip)
  int32_t bin_entry16 = 0;      // every bin spans <= 65536 vertices: offsets inside a bin fit 16-bit entries
  int32_t bin_uniform = 0;      // > 0: every bin IS the aligned range of 2^bin_uniform vertices (bin = id >> bin_uniform)
  int32_t bin_state = 0;        // 0: not built, 1: usable, 2: not applicable to this graph, 3: a column index lies outside [0, V)
  // binned relaxation of weighted SSSP (grx_relax.hpp), built lazily; owned
  int32_t* rb_off = nullptr;          // off[1025], v0[1025]
  unsigned short* rb_g2b16 = nullptr; // granule -> bin (10 bits) | index of the granule inside its bin << 10
  int32_t rb_shift = 0, rb_ngran = 0, rb_nb = 0;
  int32_t rb_uniform = 0;             // > 0: the relax bins are the aligned ranges of 2^rb_uniform vertices
  int32_t rb_state = 0;               // 0: not built, 1: usable, 2: not applicable to this graph
  std::atomic<uint32_t> rb_hint{0};   // launch groups in which a weighted search on this graph met a fat level
  void* blk[2] = {nullptr, nullptr};  // block structure of the block-asynchronous searches: [0] BFS depths, [1] weighted; owned
  int32_t blk_state[2] = {0, 0};      // 0: not tried, 1: built, 2: not applicable to this graph
  double weight_sum = -1.0;  // sum of edge weights (lazy; near-far SSSP bucket width)
  bool uniform_weights = false;
  float weight_min = 0.0f, weight_max = 0.0f;
  std::vector<int32_t> h_t_ro;  // host copy of the transpose offsets (for static partitions)
  // static PageRank pull partition (built once per graph)
  void* pr_blocks = nullptr;    // int4 {row0, nrows, e0, e1}; nrows == 0 => piece of a long row
  int32_t* pr_piece = nullptr;  // piece id per block (-1 for ordinary blocks)
  int32_t* pr_long = nullptr;   // int {row, first_piece, n_pieces} per long row
  int32_t n_pr_blocks = 0, n_pr_pieces = 0, n_pr_long = 0;
  // XCD-blocked pull layout (dense graphs): in-edges bucketed by SOURCE block so that the
  // slice of x[] a workgroup gathers from fits its XCD's L2
  int32_t* xb_ro = nullptr;     // NB * (V + 1) + 1 offsets, block-major
  int32_t* xb_ci = nullptr;
  float* xb_w = nullptr;
  void* xb_blocks = nullptr;    // int4 row blocks, the NB lists concatenated
  int32_t* xb_piece = nullptr;
  int32_t* xb_long = nullptr;   // {block-major row index, first piece, n pieces}
  int32_t* xb_perm = nullptr;   // hub-first relabelling of the gathered vector (position of v's value in x[])
  uint16_t* xb_pos = nullptr;   // per entry of xb_ci: its position inside its block BEFORE the block was sorted by source (null: unsorted)
  int32_t xb_begin[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int32_t n_xb_pieces = 0, n_xb_long = 0;
  int32_t xb_n = 8;             // source blocks of that layout (8 = one per XCD; GRX_PR_XB)
  bool has_xb = false;
};

struct grx_host_csr {
  int32_t V = 0, cols = 0, E = 0;
  int32_t directed = 1, weighted = 1, symmetric = 0;
  std::vector<int32_t> ro, ci;
  std::vector<float> w;
};
