/*
 * oracle_omp.c -- the reference's operator loop restated for ALL host cores (OpenMP).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  These are the "N-core" CPU baselines
 * bench.py reports beside the 1-core port of the reference's priority-queue oracle
 * ("an OpenMP level-synchronous BFS / parallel Bellman-Ford /
 * pull-PR on all host cores").  They follow the reference's GPU algorithm, one
 * bulk-synchronous operator pass per iteration, on host threads:
 *   BFS   include/gunrock/algorithms/bfs.hxx:105-146   advance: atomicMin(dist[n], it + 1),
 *                                                       keep if improved; filter drops the rest
 *   SSSP  include/gunrock/algorithms/sssp.hxx:116-151  relax with atomicMin on float,
 *                                                       iteration stamp de-duplicates the output
 *   PR    include/gunrock/algorithms/pr.hxx:107-152    power iteration, evaluated as a pull over
 *                                                       the transpose (no atomics), fp32
 * BFS depths / SSSP distances are the same fixed points as the reference's (checked in
 * tests/test_oracle_golden.py); they are never used as the parity yardstick.
 */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int orc_omp_threads(void) { return omp_get_max_threads(); }

/* per-thread output segments of one level, concatenated into the next frontier */
typedef struct {
  int32_t* v;
  int64_t n, cap;
} seg_t;

static void seg_push(seg_t* s, int32_t x) {
  if (s->n == s->cap) {
    s->cap = s->cap ? s->cap * 2 : 4096;
    s->v = (int32_t*)realloc(s->v, (size_t)s->cap * sizeof(int32_t));
  }
  s->v[s->n++] = x;
}

/* returns n of the concatenation written to *out (grown as needed) */
static int64_t seg_concat(seg_t* segs, int nt, int32_t** out, int64_t* out_cap) {
  int64_t total = 0;
  for (int t = 0; t < nt; ++t) total += segs[t].n;
  if (total > *out_cap) {
    *out_cap = total + total / 2 + 1024;
    *out = (int32_t*)realloc(*out, (size_t)*out_cap * sizeof(int32_t));
  }
  int64_t at = 0;
  for (int t = 0; t < nt; ++t) {
    memcpy(*out + at, segs[t].v, (size_t)segs[t].n * sizeof(int32_t));
    at += segs[t].n;
    segs[t].n = 0;
  }
  return total;
}

double orc_bfs_omp(int32_t nv, const int32_t* ro, const int32_t* ci, int32_t src, int32_t* dist,
                   int64_t* edges_visited) {
  const int nt = omp_get_max_threads();
  seg_t* segs = (seg_t*)calloc((size_t)nt, sizeof(seg_t));
  int32_t* frontier = NULL;
  int64_t cap = 0, n = 1, edges = 0;
#pragma omp parallel for schedule(static)
  for (int32_t v = 0; v < nv; ++v) dist[v] = INT32_MAX;
  cap = 1024;
  frontier = (int32_t*)malloc((size_t)cap * sizeof(int32_t));
  frontier[0] = src;
  dist[src] = 0;
  const double t0 = now_ms();
  for (int32_t it = 0; n > 0; ++it) {
    int64_t lvl_edges = 0;
#pragma omp parallel reduction(+ : lvl_edges)
    {
      seg_t* mine = &segs[omp_get_thread_num()];
#pragma omp for schedule(dynamic, 64)
      for (int64_t i = 0; i < n; ++i) {
        const int32_t v = frontier[i];
        lvl_edges += ro[v + 1] - ro[v];
        for (int32_t e = ro[v]; e < ro[v + 1]; ++e) {
          const int32_t u = ci[e];
          int32_t old = __atomic_load_n(&dist[u], __ATOMIC_RELAXED);
          while (old > it + 1) { /* atomicMin, bfs.hxx:117-119 */
            if (__atomic_compare_exchange_n(&dist[u], &old, it + 1, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
              seg_push(mine, u);
              break;
            }
          }
        }
      }
    }
    edges += lvl_edges;
    n = seg_concat(segs, nt, &frontier, &cap);
  }
  const double ms = now_ms() - t0;
  for (int t = 0; t < nt; ++t) free(segs[t].v);
  free(segs);
  free(frontier);
  if (edges_visited) *edges_visited = edges;
  return ms;
}

static inline uint32_t fbits(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
}

/* Frontier Bellman-Ford.  budget_ms > 0 stops after the iteration that exceeds it (*finished = 0:
 * distances are then NOT final -- the caller only wants the relaxation rate of the sample). */
double orc_sssp_omp(int32_t nv, const int32_t* ro, const int32_t* ci, const float* w, int32_t src, float* dist,
                    double budget_ms, int64_t* edges_relaxed, int32_t* iterations, int32_t* finished) {
  const int nt = omp_get_max_threads();
  seg_t* segs = (seg_t*)calloc((size_t)nt, sizeof(seg_t));
  int32_t* stamp = (int32_t*)malloc((size_t)nv * sizeof(int32_t));
  int64_t cap = 1024, n = 1, edges = 0;
  int32_t* frontier = (int32_t*)malloc((size_t)cap * sizeof(int32_t));
#pragma omp parallel for schedule(static)
  for (int32_t v = 0; v < nv; ++v) {
    dist[v] = FLT_MAX;
    stamp[v] = -1;
  }
  frontier[0] = src;
  dist[src] = 0.0f;
  int32_t it = 0, fin = 1;
  const double t0 = now_ms();
  for (; n > 0; ++it) {
    if (budget_ms > 0 && now_ms() - t0 > budget_ms) {
      fin = 0;
      break;
    }
    int64_t lvl_edges = 0;
#pragma omp parallel reduction(+ : lvl_edges)
    {
      seg_t* mine = &segs[omp_get_thread_num()];
#pragma omp for schedule(dynamic, 64)
      for (int64_t i = 0; i < n; ++i) {
        const int32_t v = frontier[i];
        uint32_t dvb = __atomic_load_n((uint32_t*)&dist[v], __ATOMIC_RELAXED);
        float dv;
        memcpy(&dv, &dvb, 4);
        lvl_edges += ro[v + 1] - ro[v];
        for (int32_t e = ro[v]; e < ro[v + 1]; ++e) {
          const int32_t u = ci[e];
          const float nd = dv + (w ? w[e] : 1.0f); /* sssp.hxx:121-122 */
          uint32_t old = __atomic_load_n((uint32_t*)&dist[u], __ATOMIC_RELAXED);
          /* non-negative floats order like their bit patterns */
          while (fbits(nd) < old) {
            if (__atomic_compare_exchange_n((uint32_t*)&dist[u], &old, fbits(nd), 1, __ATOMIC_RELAXED,
                                            __ATOMIC_RELAXED)) {
              /* bypass filter with the iteration stamp, sssp.hxx:132-151 */
              if (__atomic_exchange_n(&stamp[u], it, __ATOMIC_RELAXED) != it) seg_push(mine, u);
              break;
            }
          }
        }
      }
    }
    edges += lvl_edges;
    n = seg_concat(segs, nt, &frontier, &cap);
  }
  const double ms = now_ms() - t0;
  for (int t = 0; t < nt; ++t) free(segs[t].v);
  free(segs);
  free(frontier);
  free(stamp);
  if (edges_relaxed) *edges_relaxed = edges;
  if (iterations) *iterations = it;
  if (finished) *finished = fin;
  return ms;
}

/* Pull PageRank, fp32, `iterations` loop() executions exactly (no convergence test: the caller
 * times a bounded sample).  The transpose is built untimed. */
double orc_pr_omp(int32_t nv, const int32_t* ro, const int32_t* ci, const float* w, float alpha, int iterations,
                  float* p) {
  const int64_t ne = ro[nv];
  int32_t* t_ro = (int32_t*)calloc((size_t)nv + 2, sizeof(int32_t));
  int32_t* t_ci = (int32_t*)malloc((size_t)(ne > 0 ? ne : 1) * sizeof(int32_t));
  float* t_w = (float*)malloc((size_t)(ne > 0 ? ne : 1) * sizeof(float));
  float* iw = (float*)malloc((size_t)nv * sizeof(float));
  float* x = (float*)malloc((size_t)nv * sizeof(float));
  float* pn = (float*)malloc((size_t)nv * sizeof(float));
  for (int64_t e = 0; e < ne; ++e) t_ro[ci[e] + 2]++;
  for (int32_t v = 0; v < nv; ++v) t_ro[v + 2] += t_ro[v + 1];
  for (int32_t v = 0; v < nv; ++v)
    for (int32_t e = ro[v]; e < ro[v + 1]; ++e) {
      const int32_t at = t_ro[ci[e] + 1]++;
      t_ci[at] = v;
      t_w[at] = w ? w[e] : 1.0f;
    }
#pragma omp parallel for schedule(static)
  for (int32_t v = 0; v < nv; ++v) { /* pr.hxx:65-93 */
    float s = 0.0f;
    for (int32_t e = ro[v]; e < ro[v + 1]; ++e) s += w ? w[e] : 1.0f;
    iw[v] = s != 0.0f ? alpha / s : 0.0f;
    p[v] = 1.0f / (float)nv;
  }
  const double t0 = now_ms();
  for (int it = 0; it < iterations; ++it) {
    double dsum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dsum)
    for (int32_t v = 0; v < nv; ++v) { /* pr.hxx:121-132 */
      x[v] = p[v] * iw[v];
      if (iw[v] == 0.0f) dsum += (double)(alpha * p[v]);
    }
    const float base = (1.0f - alpha + (float)dsum) / (float)nv; /* pr.hxx:134 */
#pragma omp parallel for schedule(dynamic, 1024)
    for (int32_t v = 0; v < nv; ++v) { /* pr.hxx:140-146 as a gather over in-edges */
      float acc = 0.0f;
      for (int32_t e = t_ro[v]; e < t_ro[v + 1]; ++e) acc += x[t_ci[e]] * t_w[e];
      pn[v] = base + acc;
    }
    memcpy(p, pn, (size_t)nv * sizeof(float));
  }
  const double ms = now_ms() - t0;
  free(t_ro);
  free(t_ci);
  free(t_w);
  free(iw);
  free(x);
  free(pn);
  return ms;
}

/* float64 evaluation of the PageRank recurrence (pr.hxx:65-152) as a pull over the transpose on all
 * host cores, n_iter iterations exactly.  After iteration k (1-based) it records
 *   delta[k - 1]            = max |p_k - p_{k-1}|   (what is_converged compares with tol, pr.hxx:189-194)
 *   err[c * n_iter + k - 1] = max |cmp[c] - p_k|    for each of the n_cmp fp32 vectors handed in
 * so one pass yields the iteration at which the float64 recurrence converges, the distance of an fp32
 * result from the float64 iterate of ITS iteration count, and the best-matching iterate of a result
 * whose iteration count is unknown (the reference GPU path).  p_final (may be NULL) receives p_{n_iter}.
 * Differs from orc_pr_f64 only in summation order (in-edge order instead of CSR edge order). */
int orc_pr_f64_trace(int32_t nv, const int32_t* ro, const int32_t* ci, const float* w, double alpha, int n_iter,
                     int n_cmp, const float* const* cmp, double* delta, double* err, double* p_final) {
  const int64_t ne = ro[nv];
  int32_t* t_ro = (int32_t*)calloc((size_t)nv + 2, sizeof(int32_t));
  int32_t* t_ci = (int32_t*)malloc((size_t)(ne > 0 ? ne : 1) * sizeof(int32_t));
  float* t_w = (float*)malloc((size_t)(ne > 0 ? ne : 1) * sizeof(float));
  double* iw = (double*)malloc((size_t)nv * sizeof(double));
  double* x = (double*)malloc((size_t)nv * sizeof(double));
  double* p = (double*)malloc((size_t)nv * sizeof(double));
  double* pn = (double*)malloc((size_t)nv * sizeof(double));
  if (!t_ro || !t_ci || !t_w || !iw || !x || !p || !pn) return -1;
  for (int64_t e = 0; e < ne; ++e) t_ro[ci[e] + 2]++;
  for (int32_t v = 0; v < nv; ++v) t_ro[v + 2] += t_ro[v + 1];
  for (int32_t v = 0; v < nv; ++v)
    for (int32_t e = ro[v]; e < ro[v + 1]; ++e) {
      const int32_t at = t_ro[ci[e] + 1]++;
      t_ci[at] = v;
      t_w[at] = w ? w[e] : 1.0f;
    }
#pragma omp parallel for schedule(static)
  for (int32_t v = 0; v < nv; ++v) {
    double s = 0.0;
    for (int32_t e = ro[v]; e < ro[v + 1]; ++e) s += w ? (double)w[e] : 1.0;
    iw[v] = s != 0.0 ? alpha / s : 0.0;
    p[v] = 1.0 / (double)nv;
  }
  for (int it = 0; it < n_iter; ++it) {
    double dsum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dsum)
    for (int32_t v = 0; v < nv; ++v) {
      x[v] = p[v] * iw[v];
      if (iw[v] == 0.0) dsum += alpha * p[v];
    }
    const double base = (1.0 - alpha + dsum) / (double)nv;
    double dmax = 0.0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(max : dmax)
    for (int32_t v = 0; v < nv; ++v) {
      double acc = 0.0;
      for (int32_t e = t_ro[v]; e < t_ro[v + 1]; ++e) acc += x[t_ci[e]] * (double)t_w[e];
      pn[v] = base + acc;
      const double d = fabs(pn[v] - p[v]);
      if (d > dmax) dmax = d;
    }
    delta[it] = dmax;
    for (int c = 0; c < n_cmp; ++c) {
      const float* q = cmp[c];
      double emax = 0.0;
#pragma omp parallel for schedule(static) reduction(max : emax)
      for (int32_t v = 0; v < nv; ++v) {
        const double d = fabs((double)q[v] - pn[v]);
        if (d > emax) emax = d;
      }
      err[(size_t)c * (size_t)n_iter + (size_t)it] = emax;
    }
    double* t = p;
    p = pn;
    pn = t;
  }
  if (p_final) memcpy(p_final, p, (size_t)nv * sizeof(double));
  free(t_ro);
  free(t_ci);
  free(t_w);
  free(iw);
  free(x);
  free(p);
  free(pn);
  return 0;
}
