/*
 * oracle.h -- CPU restatement of the reference's BFS / SSSP / PageRank path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 * The product (libgrx.so, include/gunrock/...) never links or calls it.
 *
 * Parity status: BFS and SSSP are pinned against the reference's own CPU
 * oracle (oracle/_ref, built from /root/reference sources where they lie)
 * and against the chesapeake golden vector.
 * PageRank: "parity unpinned" by the reference (it ships no PR oracle, no
 * --validate for PR and no PR test); the restatement follows
 * include/gunrock/algorithms/pr.hxx:65-195 and is cross-checked against the
 * reference's own GPU path (oracle/_ref/libgunrock_ref_gpu.so, the reference
 * compiled here) on a GPU box: tests/test_pr_gpu.py compares ours, the
 * reference GPU result and this float64 evaluation side by side.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef GRX_ORACLE_H
#define GRX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t rows, cols, nnz;
  int32_t* row_indices;
  int32_t* column_indices;
  float* nonzero_values;
  /* graph_properties_t, include/gunrock/graph/properties.hxx:13-18 */
  int32_t directed, weighted, symmetric;
} orc_coo_t;

typedef struct {
  int32_t rows, cols, nnz;
  int32_t* row_offsets;    /* rows + 1 */
  int32_t* column_indices; /* nnz */
  float* nonzero_values;   /* nnz */
} orc_csr_t;

/* include/gunrock/io/matrix_market.hxx:99-254 (+ io/detail/mmio_impl.hxx banner
 * rules).  Returns 0 on success, non-zero error code otherwise. */
int orc_mtx_load(const char* path, orc_coo_t* out);
void orc_coo_free(orc_coo_t* coo);

/* include/gunrock/formats/csr.hxx:81-140 -- stable row bucket sort. */
int orc_csr_from_coo(const orc_coo_t* coo, orc_csr_t* out);
void orc_csr_free(orc_csr_t* csr);

/* examples/algorithms/bfs/bfs_cpu.hxx:20-68 -- priority-queue search with
 * unit edge cost; unreached = INT32_MAX.  Returns elapsed ms of the search
 * only (same timing scope as the reference). */
double orc_bfs(int32_t n_vertices,
               const int32_t* row_offsets,
               const int32_t* column_indices,
               int32_t source,
               int32_t* distances);

/* examples/algorithms/sssp/sssp_cpu.hxx:22-72 -- priority-queue Dijkstra in
 * fp32; unreached = FLT_MAX.  Returns elapsed ms. */
double orc_sssp(int32_t n_vertices,
                const int32_t* row_offsets,
                const int32_t* column_indices,
                const float* nonzero_values,
                int32_t source,
                float* distances);

/* orc_sssp with a time budget (bench.py's bounded CPU sample).  *finished = 0: distances NOT final. */
double orc_sssp_budget(int32_t n_vertices,
                       const int32_t* row_offsets,
                       const int32_t* column_indices,
                       const float* nonzero_values,
                       int32_t source,
                       float* distances,
                       double budget_ms,
                       int64_t* edges_scanned,
                       int32_t* finished);

/* include/gunrock/algorithms/pr.hxx:65-93 (reset), :107-152 (iteration),
 * :172-195 (convergence), framework/enactor.hxx:274-277 (loop order).
 * fp32 throughout, edge updates applied in CSR edge order.  Returns the
 * number of loop() executions; *elapsed_ms gets the loop time.
 * max_iterations <= 0 means unbounded (as the reference). */
int orc_pr_f32(int32_t n_vertices,
               const int32_t* row_offsets,
               const int32_t* column_indices,
               const float* nonzero_values,
               float alpha,
               float tol,
               int max_iterations,
               float* p,
               double* elapsed_ms);

/* Same recurrence evaluated in float64 (the yardstick both the reference's
 * atomics-ordered fp32 result and ours are compared to).  If
 * force_iterations > 0 the convergence test is skipped and exactly that many
 * iterations run. */
int orc_pr_f64(int32_t n_vertices,
               const int32_t* row_offsets,
               const int32_t* column_indices,
               const float* nonzero_values,
               double alpha,
               double tol,
               int max_iterations,
               int force_iterations,
               double* p,
               double* elapsed_ms);

/* Size-independent property checkers used at BASELINE.json's full sizes
 * (exact characterisation of the fixed points; see DESIGN.md "Parity").
 * Return the number of violations. */
int64_t orc_check_bfs(int32_t n_vertices,
                      const int32_t* row_offsets,
                      const int32_t* column_indices,
                      int32_t source,
                      const int32_t* distances);
int64_t orc_check_sssp(int32_t n_vertices,
                       const int32_t* row_offsets,
                       const int32_t* column_indices,
                       const float* nonzero_values,
                       int32_t source,
                       const float* distances);

/* Plain queue BFS (not the reference's algorithm; same depths).  Used only to
 * cross-check orc_bfs and as a fast checker on large graphs. */
double orc_bfs_queue(int32_t n_vertices,
                     const int32_t* row_offsets,
                     const int32_t* column_indices,
                     int32_t source,
                     int32_t* distances,
                     int64_t* edges_visited);

/* ---- N-core baselines (oracle_omp.c): the reference's bulk-synchronous operator loop on all
 * host cores.  Same fixed points as the functions above; used by bench.py's cpu_baseline leg and
 * cross-checked in tests, never as the parity yardstick. ---- */
int orc_omp_threads(void);
/* include/gunrock/algorithms/bfs.hxx:105-146 per level on host threads */
double orc_bfs_omp(int32_t n_vertices, const int32_t* row_offsets, const int32_t* column_indices,
                   int32_t source, int32_t* distances, int64_t* edges_visited);
/* include/gunrock/algorithms/sssp.hxx:116-151 per iteration on host threads.  budget_ms > 0: stop
 * after the iteration that exceeds it (*finished = 0, distances not final: rate sample only). */
double orc_sssp_omp(int32_t n_vertices, const int32_t* row_offsets, const int32_t* column_indices,
                    const float* nonzero_values, int32_t source, float* distances, double budget_ms,
                    int64_t* edges_relaxed, int32_t* iterations, int32_t* finished);
/* include/gunrock/algorithms/pr.hxx:107-152 as a pull over the transpose, exactly `iterations`
 * loop() executions, fp32.  Returns the loop time in ms. */
double orc_pr_omp(int32_t n_vertices, const int32_t* row_offsets, const int32_t* column_indices,
                  const float* nonzero_values, float alpha, int iterations, float* p);

/* float64 pull evaluation of the PageRank recurrence with a per-iteration trace: see oracle_omp.c.
 * delta[n_iter], err[n_cmp * n_iter]; returns 0, or -1 when out of memory. */
int orc_pr_f64_trace(int32_t n_vertices, const int32_t* row_offsets, const int32_t* column_indices,
                     const float* nonzero_values, double alpha, int n_iter, int n_cmp,
                     const float* const* cmp, double* delta, double* err, double* p_final);

#ifdef __cplusplus
}
#endif
#endif
