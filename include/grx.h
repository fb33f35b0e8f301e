/*
 * grx.h -- C ABI of the MI355X-native frontier engine (libgrx.so).
 *
 * This is the drop-in boundary for the hot path: every entry point below is
 * what a foreign-function binding of the reference's BFS / SSSP / PageRank
 * path would bind.  The reference (gunrock "essentials") has no C ABI of its
 * own; its public surfaces are C++ templates, a CLI and a nanobind module that
 * passes torch `data_ptr()` device pointers.  Each function cites the
 * reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types;
 *   - `d_` pointers are DEVICE memory owned by the caller (the reference's
 *     result buffers are caller-allocated device memory too:
 *     include/gunrock/algorithms/bfs.hxx:185-189);
 *   - every call returns a grx_status_t; grx_last_error_string() gives the
 *     message (the reference throws gunrock::error::exception_t,
 *     include/gunrock/error.hxx:20-45);
 *   - vertex_t = edge_t = int32, weight_t = float32, as every reference driver
 *     and binding instantiates (examples/algorithms/bfs/bfs.cu:15-17,
 *     python/src/gunrock/bindings.cu:49-51);
 *   - elapsed times are milliseconds measured with two events on the
 *     context's stream around the enact loop, reset excluded -- the
 *     reference's timing scope (include/gunrock/framework/enactor.hxx:270-282).
 */
#ifndef GRX_H
#define GRX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GRX_SUCCESS = 0,
  GRX_ERROR_INVALID_ARGUMENT = 1,
  GRX_ERROR_HIP = 2,            /* a HIP runtime call failed */
  GRX_ERROR_UNSUPPORTED = 3,    /* option combination not supported */
  GRX_ERROR_IO = 4,             /* file could not be read / parsed */
  GRX_ERROR_OUT_OF_MEMORY = 5,
  GRX_ERROR_NOT_CONVERGED = 6
} grx_status_t;

/* operators::load_balance_t, include/gunrock/framework/operators/configs.hxx:52-60
 * (same integer values; the Python module of the reference exposes the ints). */
typedef enum {
  GRX_LB_THREAD_MAPPED = 0,
  GRX_LB_WARP_MAPPED = 1,   /* enum-only in the reference; real here (64-lane wave per vertex) */
  GRX_LB_BLOCK_MAPPED = 2,  /* reference default */
  GRX_LB_BUCKETING = 3,     /* stub in the reference; mapped to the degree-binned selector */
  GRX_LB_MERGE_PATH = 4,
  GRX_LB_MERGE_PATH_V2 = 5, /* NVIDIA-only in the reference; same kernel as MERGE_PATH here */
  GRX_LB_WORK_STEALING = 6  /* enum-only in the reference; GRX_ERROR_UNSUPPORTED */
} grx_load_balance_t;

/* operators::filter_algorithm_t, configs.hxx:93-98 */
typedef enum {
  GRX_FILTER_REMOVE = 0,
  GRX_FILTER_PREDICATED = 1,
  GRX_FILTER_COMPACT = 2,   /* throws in the reference (filter/compact.hxx:21-24); real here */
  GRX_FILTER_BYPASS = 3
} grx_filter_algorithm_t;

/* operators::uniquify_algorithm_t, configs.hxx:100-105 */
typedef enum { GRX_UNIQUIFY_UNIQUE = 0, GRX_UNIQUIFY_UNIQUE_COPY = 1 } grx_uniquify_algorithm_t;

/* operators::advance_direction_t, configs.hxx:78-82 */
typedef enum { GRX_DIR_FORWARD = 0, GRX_DIR_BACKWARD = 1, GRX_DIR_OPTIMIZED = 2 } grx_direction_t;

/* gunrock::options_t, include/gunrock/algorithms/algorithms.hxx:27-72 --
 * the first seven fields mirror it field for field with the same defaults
 * (grx_options_default).  The trailing fields are engine extensions; zero
 * means "engine decides" so a zero-initialised tail keeps reference behaviour. */
typedef struct grx_options {
  int32_t advance_load_balance; /* grx_load_balance_t, default BLOCK_MAPPED */
  int32_t filter_algorithm;     /* grx_filter_algorithm_t, default PREDICATED */
  int32_t enable_filter;        /* default 0 */
  int32_t enable_uniquify;      /* default 0 */
  int32_t uniquify_algorithm;   /* default UNIQUE */
  int32_t best_effort_uniquify; /* default 1 */
  float uniquify_percent;       /* default 100 */
  /* --- extensions --- */
  int32_t engine_flags;         /* GRX_FLAG_* bitmask */
  int32_t advance_direction;    /* grx_direction_t (ignored by the reference) */
  int32_t max_iterations;       /* 0 = unbounded, as the reference */
  int32_t reserved[5];
} grx_options_t;

/* engine_flags */
#define GRX_FLAG_UNFUSED 0x1   /* run advance and filter as separate operators exactly as the
                                  reference pipeline does (frontier with -1 holes between them) */
#define GRX_FLAG_PROFILE 0x2   /* record per-iteration operator timings (adds events + syncs) */
#define GRX_FLAG_SYNC_EACH_LEVEL 0x4 /* host reads the frontier size after every level */
#define GRX_FLAG_LB_STRICT 0x1000    /* advance_load_balance is an ORDER, not a hint.  By default the engine picks the
                                        body of every level on the device from the frontier's statistics (vertices, out-edges,
                                        mean degree, share of the graph already visited): LDS-resident tiny levels, many
                                        mid-size levels per launch (block-mapped staging), the chunked merge-path advance,
                                        binned scatter + claim, bottom-up; grx_level_profile_t::body reports the choice.
                                        With this flag and MERGE_PATH every level runs the chunked merge-path advance (the
                                        reference pipeline of BASELINE configs[1] as written); with BLOCK_MAPPED the
                                        block-staged bodies are preferred wherever the frontier fits them */
#define GRX_FLAG_NO_BLOCK_ASYNC 0x2000 /* with GRX_BLOCK=1 (opt-in): keep the level-synchronous kernels for this call (see grx_get_block_stats) */
#define GRX_FLAG_ASYNC_RETURN 0x8    /* grx_bfs may return as soon as the device has PUBLISHED the end
                                        of the search (results final) instead of after its stream drained;
                                        see grx_bfs.  Off by default: the reference's run() returns after a
                                        stream synchronisation (enactor.hxx:280-282) */

/* bits 0x10 .. 0x80 are per algorithm (tests / A-B runs): */
#define GRX_FLAG_PR_NO_XCD_LAYOUT 0x40 /* grx_pr: never use the XCD-blocked copy of the in-edges */
#define GRX_FLAG_PR_XCD_LAYOUT 0x80    /* grx_pr: always use it */
#define GRX_FLAG_SSSP_PLAIN 0x10     /* grx_sssp: label-correcting levels only, no near-far (delta-stepping) buckets */
#define GRX_FLAG_SSSP_NEAR_FAR 0x20  /* grx_sssp: near-far buckets also on dense graphs (mean degree >= 6) */
#define GRX_FLAG_SSSP_NO_BFS 0x40    /* grx_sssp: relax with the SSSP kernels even when all weights are equal.  By default
                                        such a graph (every pattern .mtx: the reference loader stores 1.0, io/matrix_market.hxx:
                                        170-171) is searched by the BFS engine and the depths become k-fold fp32 sums of w */
#define GRX_FLAG_SSSP_NO_BINS 0x80   /* grx_sssp: every level on the relax-per-edge advance.  By default the fat levels of a
                                        weighted search on a dense graph run as a binned relaxation: (target, tentative distance)
                                        pairs scattered to bins by target range, minimum taken in LDS (grx_relax.hpp) */

typedef struct grx_context* grx_context_t;
typedef struct grx_graph* grx_graph_t;

void grx_options_default(grx_options_t* options);

const char* grx_last_error_string(void);
const char* grx_version_string(void);

/* gcuda::multi_context_t(device) / standard_context_t: device ordinal, one
 * stream, an event pair (include/gunrock/cuda/context.hxx:54-216).  `stream`
 * may be an existing hipStream_t (e.g. torch's current stream) or NULL to
 * create a non-blocking one as the reference does (:84). */
grx_status_t grx_context_create(int32_t device, void* stream, grx_context_t* out);
grx_status_t grx_context_synchronize(grx_context_t ctx); /* context.hxx:124-128 */
grx_status_t grx_context_destroy(grx_context_t ctx);
void* grx_context_stream(grx_context_t ctx);
/* Order the context's stream after the work queued so far on `producer_stream` (another stream of the same device, e.g.
 * the one a caller filled an input array on) without blocking the host: nothing is done when that stream is the
 * context's own or already idle, else an event is recorded there and waited for on the context's stream.  The reference
 * has no counterpart (its algorithms run on the stream the caller's thrust calls use). */
grx_status_t grx_context_order_after(grx_context_t ctx, void* producer_stream);

/* graph::build<memory_space_t::device>(properties, csr),
 * include/gunrock/graph/build.hxx:29-36 + graph/csr.hxx:218-226: a NON-OWNING
 * view over the caller's device CSR arrays.  d_values may be NULL (treated as
 * all 1.0, what the reference loader produces for pattern files,
 * io/matrix_market.hxx:170-171).
 * `symmetric` (graph_properties_t::symmetric, default true and inert in the reference) is
 * load-bearing here -- a symmetric graph's CSR serves as its in-edge list in the bottom-up
 * BFS step -- and therefore VERIFIED on the device the first time it is relied on (one pass
 * over the edges, cached in the handle); a CSR that is not its own transpose gets a real
 * transpose instead. */
grx_status_t grx_graph_create_csr(grx_context_t ctx,
                                  int32_t n_vertices,
                                  int32_t n_edges,
                                  const int32_t* d_row_offsets,
                                  const int32_t* d_column_indices,
                                  const float* d_values,
                                  int32_t directed,
                                  int32_t weighted,
                                  int32_t symmetric,
                                  grx_graph_t* out);
grx_status_t grx_graph_destroy(grx_graph_t graph);
/* A graph handle caches per-graph derived state (transpose, pull partitions, weight statistics),
 * so the CSR arrays must not change under a live handle.  A caller that keys handles on array
 * identity -- the C++ bridge include/gunrock/algorithms/engine.hxx does, because the reference's
 * graph_t is a non-owning by-value view (graph/graph.hxx:187-214) -- compares this SAMPLED content
 * fingerprint (1024 positions of each array + the edge count; one tiny kernel + one 8-byte
 * read-back) to detect in-place edits or a different graph allocated at the same addresses. */
grx_status_t grx_csr_fingerprint(grx_context_t ctx, int32_t n_vertices, int32_t n_edges,
                                 const int32_t* d_row_offsets, const int32_t* d_column_indices,
                                 const float* d_values, uint64_t* out);
/* The FULL content hash of the three arrays (every element, position-keyed; one streaming pass at HBM rate, one
 * 8-byte read-back and stream synchronisation): what a caller that must tolerate in-place edits compares per use.
 * include/gunrock/algorithms/engine.hxx does by default, because upstream's graph view is non-owning and editing the
 * arrays between two run() calls is legal there (graph/graph.hxx:187-214). */
grx_status_t grx_csr_hash(grx_context_t ctx, int32_t n_vertices, int32_t n_edges,
                          const int32_t* d_row_offsets, const int32_t* d_column_indices,
                          const float* d_values, uint64_t* out);
int32_t grx_graph_number_of_vertices(grx_graph_t graph); /* graph_t::get_number_of_vertices */
int32_t grx_graph_number_of_edges(grx_graph_t graph);    /* graph_t::get_number_of_edges */

/* float gunrock::bfs::run(G, param, result, context),
 * include/gunrock/algorithms/bfs.hxx:162-182 (legacy overload :201-215).
 * d_distances[V] int32: depth, INT32_MAX if unreached (bfs.hxx:65-66).
 * d_predecessors may be NULL; like the reference it is accepted and never
 * written (bfs.hxx:29).
 * *elapsed_ms: enact() time, seed -> convergence, reset excluded (enactor.hxx:270-282).
 * Completion: on return d_distances is final and the context's stream has drained, like the
 * reference's run().  With GRX_FLAG_ASYNC_RETURN (opt-in; scale-free graphs, E >= 8 V) the call
 * returns as soon as the device publishes the end of the search in pinned host memory -- the
 * distances are final, but at most three no-op kernel groups may still be draining on the context's
 * stream (work enqueued on that stream afterwards is ordered behind them,
 * grx_context_synchronize waits for them). */
grx_status_t grx_bfs(grx_context_t ctx,
                     grx_graph_t graph,
                     int32_t single_source,
                     const grx_options_t* options, /* NULL = defaults */
                     int32_t* d_distances,
                     int32_t* d_predecessors,
                     float* elapsed_ms);

/* float gunrock::sssp::run(G, param, result, context),
 * include/gunrock/algorithms/sssp.hxx:176-198.  d_distances[V] float32,
 * FLT_MAX if unreached (sssp.hxx:72-73). */
grx_status_t grx_sssp(grx_context_t ctx,
                      grx_graph_t graph,
                      int32_t single_source,
                      const grx_options_t* options,
                      float* d_distances,
                      int32_t* d_predecessors,
                      float* elapsed_ms);

/* float gunrock::pr::run(G, param, result, context),
 * include/gunrock/algorithms/pr.hxx:211-236.  d_p[V] float32 ranks.
 * iterations (may be NULL) receives the number of loop() executions. */
grx_status_t grx_pr(grx_context_t ctx,
                    grx_graph_t graph,
                    float alpha,
                    float tol,
                    const grx_options_t* options,
                    float* d_p,
                    int32_t* iterations,
                    float* elapsed_ms);

/* Run statistics of the last algorithm call on this context: what the
 * reference's metrics build collects (include/gunrock/framework/benchmark.hxx:50-60,
 * include/gunrock/util/performance.hxx:225-229 -- mteps = edges_visited /
 * runtime_ms / 1000). */
typedef struct grx_run_stats {
  int64_t edges_visited;     /* sum over iterations of degrees of valid frontier vertices */
  int64_t vertices_visited;  /* valid frontier vertices summed over iterations */
  int32_t search_depth;      /* enactor iterations executed */
  int32_t n_levels_recorded; /* entries valid in the per-level arrays (GRX_FLAG_PROFILE) */
  float elapsed_ms;
  float reserved;
} grx_run_stats_t;
grx_status_t grx_get_run_stats(grx_context_t ctx, grx_run_stats_t* out);

/* OPT-IN (environment GRX_BLOCK=1; the default is the level-synchronous kernels, which measured as fast or faster --
 * DESIGN.md, "block-asynchronous relaxation"): road-like graphs (fewer than 4 edges per vertex, >= 65536 vertices) are then
 * searched BLOCK-ASYNCHRONOUSLY (gunrock_amd/csrc/grx_block.hip; GRX_FLAG_NO_BLOCK_ASYNC keeps the level-synchronous kernels
 * for one call even with GRX_BLOCK=1, and does nothing without it): blocks of a few
 * thousand vertices relax to their local fixed point in LDS, supersteps synchronise only between blocks, global buckets
 * keep wrong labels from flooding.  Statistics of the last such search on the context (supersteps == 0: the last search
 * took another path).  For these searches grx_run_stats_t::edges_visited is the reference's count for BFS (out-edges of
 * the reached vertices) and the edges relaxed for weighted SSSP; edges_relaxed below always counts every relaxation. */
typedef struct grx_block_stats {
  int64_t edges_relaxed;   /* intra-block + boundary relaxations, re-relaxations included */
  int64_t activations;     /* block activations summed over the supersteps */
  int64_t cross_edges;     /* edges of the graph that leave their block */
  int32_t supersteps;      /* head + block launch pairs that had work */
  int32_t buckets;         /* global label buckets opened */
  int32_t blocks;          /* blocks the graph was cut into ... */
  int32_t block_vertices;  /* ... of at most this many vertices */
  double build_ms;         /* one-time cost of the block structure of this graph (host partitioner + upload) */
} grx_block_stats_t;
grx_status_t grx_get_block_stats(grx_context_t ctx, grx_block_stats_t* out);
/* 1: this build carries that path (gunrock_amd/libgrx_block.so, `python -m gunrock_amd.build --with-block`), 0: it does not -- the
 * default library since round 6: the path is opt-in and measured no better than the level-synchronous kernels */
int32_t grx_has_block_async(void);

/* HOST emulation of that schedule on a host CSR (the partitioner, the block structure and the superstep / bucket / local
 * round logic of grx_block.hip, executed serially): test infrastructure of the CPU suite, never called by a product path.
 * out_keys[V]: depths (weighted == 0; INT32_MAX unreached) or the bit patterns of float distances (FLT_MAX unreached);
 * delta_bits: bucket width in hops, or the bits of a float. */
typedef struct grx_host_csr* grx_host_csr_t;
grx_status_t grx_debug_block_search_host(grx_host_csr_t csr, int32_t weighted, int32_t block_vertices, int32_t source,
                                         uint32_t delta_bits, uint32_t* out_keys, grx_block_stats_t* stats);

/* Per-level profile of the last run with GRX_FLAG_PROFILE: up to `capacity`
 * levels are copied; fields below. */
typedef struct grx_level_profile {
  int64_t frontier_size;  /* valid vertices of the input frontier */
  int64_t edges;          /* their out-degree sum (the level's traversed edges) */
  float advance_ms;       /* advance (top-down) or bottom-up kernel time */
  float other_ms;         /* planning / direction / conversion kernels */
  int32_t bottom_up;      /* body the device chose for the level: 0 chunked merge-path advance, 1 bottom-up
                             (direction-optimising BFS), 2 binned scatter + claim (grx_bin.hpp; near-far SSSP: a
                             bucket pull), 3 many mid-size levels in one launch (grx_mid.hpp: the record then covers
                             all of them) */
  int32_t reserved;
  int64_t bu_open;        /* bottom-up: unvisited vertices examined */
  int64_t bu_probes;      /* bottom-up: in-edges read */
} grx_level_profile_t;
grx_status_t grx_get_level_profile(grx_context_t ctx, grx_level_profile_t* out,
                                   int32_t capacity, int32_t* n_levels);

/* Test hook: the stable LSD radix sort behind the per-graph preprocessing (gunrock_amd/csrc/grx_sort.hpp) on caller
 * device arrays, in place: n (key, value[, value2]) triples ordered by the low key_bits bits of the key, ties in input order. */
grx_status_t grx_debug_radix_sort(grx_context_t ctx, uint32_t* d_keys, uint32_t* d_vals, uint32_t* d_vals2_or_null, int64_t n,
                                  int32_t key_bits);
/* Tuning aid: copies `n` 64-bit words of the context's debug scratch (per-workgroup timeline of the binned BFS
 * kernels, recorded when GRX_BIN_DEBUG=<level> is set; tools/bin_debug.py). */
grx_status_t grx_debug_read(grx_context_t ctx, long long* out, int64_t n);
/* Tuning aid: the first n (<= 5) spare counters of the device control block (GRX_MID_DEBUG=1: wall-clock ticks the
 * leader of a multi-level launch spent per phase, and the levels it ran; tools/history/ab_mid.py). */
grx_status_t grx_debug_ctrl(grx_context_t ctx, int32_t* out, int32_t n);

/* ---- multi-GPU: level-group interface of the partitioned BFS enactor ------------------
 * The reference has no multi-GPU execution (every operator throws when
 * context.size() != 1: framework/operators/advance/advance.hxx:129-132).  One process
 * per GPU drives these calls and runs the two collectives of a level group with RCCL
 * (inside the library after grx_bfs_dist_comm_init, or torch.distributed: all_to_all_single
 * of fixed-size bitmaps, all_reduce of 4 words); see gunrock_amd/distributed.py, the second
 * half of gunrock_amd/csrc/grx_bfs.hip and DESIGN.md section 7.
 *
 * A rank runs the single-GPU ENGINE on the rows it owns (round 6): the same search object, the same kernels --
 * binned scatter + sweep on fat levels, the second bottom-up body, the claim-per-edge advance on thin levels -- with
 * the partition's rule for targets of other ranks (reported once through the outgoing bitmap) and one kernel behind the
 * exchange that claims what the peers reported.  With ONE rank a partitioned search IS grx_bfs.
 *
 * Partition: S = grx_bfs_dist_slice_bits(V, n_ranks) (a multiple of 2048); rank r owns
 * vertices [r * S, min((r + 1) * S, V)).  `out_rows` holds the CSR rows of the owned
 * vertices with GLOBAL column ids (n_vertices = global V, other rows empty); `in_rows`
 * the same for the in-edges (NULL: the graph is symmetric, or no bottom-up step).
 * Caller-owned device buffers (torch tensors in the Python driver):
 *   d_send, d_recv : parts * n_ranks * (S / 32) 32-bit words each (zero-initialised)
 *   d_stats_local, d_stats_global : int64[4]
 * Per level group the caller enqueues, all asynchronously on the context's stream:
 *   grx_bfs_dist_pre(h, 0); all_to_all_single(recv[0] <- send[0]);
 *   [parts == 2, kept for callers of the two-halves protocol: grx_bfs_dist_pre(h, 1) is a no-op and half 1 of the
 *    buffers stays empty -- every report of a level travels in half 0]
 *   grx_bfs_dist_post(h); all_reduce(stats_global <- stats_local)
 * -- blindly, several groups per grx_bfs_dist_poll; groups after `done` are no-ops.
 * Or, one call per search: grx_bfs_dist_run (below). */
typedef struct grx_bfs_dist* grx_bfs_dist_t;
int32_t grx_bfs_dist_slice_bits(int32_t n_vertices, int32_t n_ranks);
grx_status_t grx_bfs_dist_create(grx_context_t ctx, grx_graph_t out_rows, grx_graph_t in_rows_or_null,
                                 int32_t n_ranks, int32_t my_rank, long long n_edges_global, int32_t parts,
                                 void* d_send, void* d_recv, long long* d_stats_local,
                                 const long long* d_stats_global, grx_bfs_dist_t* out);
/* reset + seed; afterwards the caller all-reduces stats_local into stats_global once.
 * advance_direction: GRX_DIR_FORWARD (top-down only) or GRX_DIR_OPTIMIZED.
 * d_distances: device int32[V]; only the owned range is written (and authoritative). */
grx_status_t grx_bfs_dist_begin(grx_bfs_dist_t h, int32_t source, int32_t advance_direction,
                                int32_t* d_distances);
/* the same with SHARDED labels: d_local holds only the owned slice, S = grx_bfs_dist_slice_bits(V, n_ranks)
 * entries; vertex v of this rank is at d_local[v - my_rank * S] */
grx_status_t grx_bfs_dist_begin_local(grx_bfs_dist_t h, int32_t source, int32_t advance_direction,
                                      int32_t* d_local);
grx_status_t grx_bfs_dist_pre(grx_bfs_dist_t h, int32_t part);
grx_status_t grx_bfs_dist_post(grx_bfs_dist_t h);
/* synchronises the stream; *done != 0 once the global frontier ran empty */
grx_status_t grx_bfs_dist_poll(grx_bfs_dist_t h, int32_t* done, int32_t* level);
/* run statistics: edges / vertices are this rank's share, search_depth is global */
grx_status_t grx_bfs_dist_end(grx_bfs_dist_t h, grx_run_stats_t* stats);
grx_status_t grx_bfs_dist_destroy(grx_bfs_dist_t h);

/* ---- multi-GPU: partitioned PageRank -------------------------------------------------------
 * Rank r owns the vertices [lo, hi): `out_rows` / `in_rows` are V-row CSRs holding the out-edges / in-edges of the
 * owned vertices only (global column ids; grx_host_csr_generate_rows / _in_rows produce exactly these).  p is SHARDED
 * (d_p_local: hi - lo floats).  One iteration = grx_pr_dist_pre (x = p * iweights for the owned rows into this rank's
 * slice of the global x buffer; the rank's {dangling sum, norm} pair), the caller's two all-gathers (x slices, pairs:
 * torch.distributed = RCCL in gunrock_amd/distributed.py), grx_pr_dist_post (convergence test and base term from
 * the gathered pairs -- identical on every rank -- then the pull over the owned rows).  Iterations are enqueued blindly,
 * several per grx_pr_dist_poll; iterations after `done` are no-ops.  The recurrence is the reference's
 * (algorithms/pr.hxx:107-195); the reference itself has no multi-GPU execution. */
typedef struct grx_pr_dist* grx_pr_dist_t;
grx_status_t grx_pr_dist_create(grx_context_t ctx, grx_graph_t out_rows, grx_graph_t in_rows, int32_t n_ranks,
                                int32_t my_rank, int32_t lo, int32_t hi, grx_pr_dist_t* out);
/* d_x: the global x buffer (>= V floats; this rank writes [lo, hi)); d_pair_out: this rank's 2 words;
 * d_pairs: 2 * n_ranks words, the all-gathered pairs */
grx_status_t grx_pr_dist_begin(grx_pr_dist_t h, float alpha, float tol, float* d_p_local, float* d_x,
                               uint32_t* d_pair_out, const uint32_t* d_pairs);
grx_status_t grx_pr_dist_pre(grx_pr_dist_t h);
grx_status_t grx_pr_dist_post(grx_pr_dist_t h);
grx_status_t grx_pr_dist_poll(grx_pr_dist_t h, int32_t* done, int32_t* iterations);
grx_status_t grx_pr_dist_end(grx_pr_dist_t h, grx_run_stats_t* stats);
grx_status_t grx_pr_dist_destroy(grx_pr_dist_t h);

/* ---- multi-GPU: partitioned SSSP ----------------------------------------------------------
 * Rank r owns the vertex slice [r * S, min((r + 1) * S, V)), S = grx_bfs_dist_slice_bits(V, n_ranks), and its out-rows
 * (a V-row CSR with the other rows empty, global column ids); labels are SHARDED (d_local: S floats).  One
 * iteration group = grx_sssp_dist_pre (termination from the all-reduced frontier size, chunk map, advance: owned
 * targets are relaxed on the label, remote ones min-reduced into d_send -- n_ranks slices of S floats, slice j for
 * rank j), the caller's all_to_all_single of the slices (d_send -> d_recv) , grx_sssp_dist_post (the owner applies the
 * minima it received and appends the improved vertices to its frontier; size of the next frontier into
 * d_stats_local[0]), the caller's all-reduce of d_stats_local into d_stats_global.  Groups are enqueued blindly,
 * several per grx_sssp_dist_poll; groups after `done` are no-ops.  Distances equal grx_sssp's bit for bit.  The
 * recurrence is the reference's (algorithms/sssp.hxx:104-159); the reference has no multi-GPU execution. */
typedef struct grx_sssp_dist* grx_sssp_dist_t;
grx_status_t grx_sssp_dist_create(grx_context_t ctx, grx_graph_t out_rows, int32_t n_ranks, int32_t my_rank,
                                  float* d_send, const float* d_recv, long long* d_stats_local,
                                  const long long* d_stats_global, grx_sssp_dist_t* out);
grx_status_t grx_sssp_dist_begin(grx_sssp_dist_t h, int32_t source, float* d_local);
grx_status_t grx_sssp_dist_pre(grx_sssp_dist_t h);
grx_status_t grx_sssp_dist_post(grx_sssp_dist_t h);
grx_status_t grx_sssp_dist_poll(grx_sssp_dist_t h, int32_t* done, int32_t* iteration);
grx_status_t grx_sssp_dist_end(grx_sssp_dist_t h, grx_run_stats_t* stats);
grx_status_t grx_sssp_dist_destroy(grx_sssp_dist_t h);

/* RCCL transport INSIDE the library (opt-in; librccl is opened at run time): after grx_bfs_dist_comm_init the
 * two collectives of a level group are issued from C on the context's stream -- a grouped ncclSend/ncclRecv per
 * peer for the bitmaps (each pair of GPUs has its own xGMI link) and an ncclAllReduce of the 4 statistics words
 * -- so the host makes ONE call per batch of level groups and a group can be captured into a HIP graph without
 * any Python in between.
 *   rank 0: grx_dist_unique_id(buf) (grx_dist_unique_id_bytes() bytes), broadcast to all ranks by any means;
 *   all   : grx_bfs_dist_comm_init(h, buf)            ncclCommInitRank(n_ranks, id, my_rank)
 *   search: grx_bfs_dist_begin[_local]; grx_bfs_dist_seed_stats; { grx_bfs_dist_groups(h, n); grx_bfs_dist_poll }
 *           until done; grx_bfs_dist_end; once: grx_bfs_dist_capture_group (records one group; later calls of
 *           grx_bfs_dist_groups with the same label buffer / direction replay it; a failed capture is remembered
 *           and the eager path stays). */
int32_t grx_dist_unique_id_bytes(void);
grx_status_t grx_dist_unique_id(void* out);
grx_status_t grx_bfs_dist_comm_init(grx_bfs_dist_t h, const void* unique_id);
grx_status_t grx_bfs_dist_seed_stats(grx_bfs_dist_t h);
grx_status_t grx_bfs_dist_groups(grx_bfs_dist_t h, int32_t n);
grx_status_t grx_bfs_dist_capture_group(grx_bfs_dist_t h);
int32_t grx_bfs_dist_group_is_captured(grx_bfs_dist_t h);
/* A whole partitioned search in ONE call.  n_ranks == 1: exactly grx_bfs (same search object, kernels and launch
 * schedule).  n_ranks > 1: after grx_bfs_dist_comm_init -- begin, the seed's all-reduce, batches of level groups (the first
 * as long as the previous search on the handle) with one look at `done` per batch, end; every rank must call it alike.
 * labels_are_local != 0: d_labels holds the owned slice only (as grx_bfs_dist_begin_local), else V entries of which the
 * owned range is written.  stats: edges / vertices are this rank's share, search_depth is global. */
grx_status_t grx_bfs_dist_run(grx_bfs_dist_t h, int32_t source, int32_t advance_direction, int32_t* d_labels,
                              int32_t labels_are_local, grx_run_stats_t* stats);

/* ---- host-side ingest (same semantics as the reference) ---- */

/* io::matrix_market_t::load + format::csr_t::from_coo,
 * include/gunrock/io/matrix_market.hxx:99-254, include/gunrock/formats/csr.hxx:81-140.
 * Produces host CSR arrays owned by the returned handle. */
grx_status_t grx_host_csr_load_mtx(const char* filename, grx_host_csr_t* out);
/* format::csr_t::read_binary / write_binary, formats/csr.hxx:142-228 */
grx_status_t grx_host_csr_read_binary(const char* filename, grx_host_csr_t* out);
grx_status_t grx_host_csr_write_binary(grx_host_csr_t csr, const char* filename);
/* COO (host arrays, caller-owned) -> CSR with the reference's stable row
 * bucket sort (formats/csr.hxx:81-140). */
grx_status_t grx_host_csr_from_coo(int32_t n_rows, int32_t n_cols, int64_t nnz,
                                   const int32_t* row_indices,
                                   const int32_t* column_indices,
                                   const float* values, /* NULL = 1.0 */
                                   grx_host_csr_t* out);
/* The same conversion on the DEVICE (device arrays, caller-owned, on the context's device): stable by row -- the entries of a
 * row keep their input order, duplicates and self loops included, exactly as the host counting sort leaves them
 * (formats/csr.hxx:81-140) -- so the result is byte-identical to grx_host_csr_from_coo on the same triples.
 * d_row_offsets: n_rows + 1 ints; d_out_columns / d_out_values: nnz entries; d_values and d_out_values both NULL for a
 * pattern.  Synchronous; a row index outside [0, n_rows) is an error. */
grx_status_t grx_csr_from_coo_device(grx_context_t ctx, int32_t n_rows, int64_t nnz,
                                     const int32_t* d_row_indices, const int32_t* d_column_indices,
                                     const float* d_values, int32_t* d_row_offsets,
                                     int32_t* d_out_columns, float* d_out_values);
grx_status_t grx_host_csr_info(grx_host_csr_t csr, int32_t* n_vertices, int32_t* n_edges,
                               int32_t* directed, int32_t* weighted, int32_t* symmetric);
const int32_t* grx_host_csr_row_offsets(grx_host_csr_t csr);
const int32_t* grx_host_csr_column_indices(grx_host_csr_t csr);
const float* grx_host_csr_values(grx_host_csr_t csr);
grx_status_t grx_host_csr_destroy(grx_host_csr_t csr);

/* Seeded synthetic stand-ins for the BASELINE.json graphs;
 * counter-based generator, identical output for identical arguments on any
 * machine.  kind: 0 = R-MAT(a,b,c) directed pattern, 1 = R-MAT symmetric
 * pattern (Kronecker-style, entries doubled as the loader does), 2 = road-like
 * lattice symmetric (p_keep = a; weighted with integer weights 1..1000 if
 * c > 0, else unit weights). */
grx_status_t grx_host_csr_generate(int32_t kind, int32_t n_vertices, int64_t n_entries,
                                   float a, float b, float c, uint64_t seed,
                                   grx_host_csr_t* out);
/* Same generator, keeping only the rows [row_lo, row_hi) (the slice a rank owns);
 * the result still has n_vertices rows (the others empty) and global column ids.
 * kinds 0 and 1 only. */
grx_status_t grx_host_csr_generate_rows(int32_t kind, int32_t n_vertices, int64_t n_entries,
                                        float a, float b, float c, uint64_t seed,
                                        int32_t row_lo, int32_t row_hi, grx_host_csr_t* out);
/* The IN-rows [row_lo, row_hi) of the same graph (rows of its transpose, global source
 * ids): what a rank needs for the bottom-up step of a directed graph. */
grx_status_t grx_host_csr_generate_in_rows(int32_t kind, int32_t n_vertices, int64_t n_entries,
                                           float a, float b, float c, uint64_t seed,
                                           int32_t row_lo, int32_t row_hi, grx_host_csr_t* out);

#ifdef __cplusplus
}
#endif
#endif /* GRX_H */
