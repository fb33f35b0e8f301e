/*
 * oracle.c -- CPU restatement of the reference's BFS / SSSP / PageRank path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h for the rules and parity status).
 * Plain C99, single-threaded like the reference's own CPU path.
 * Citations are paths relative to /root/reference.
 */
#include "oracle.h"

#include <ctype.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------------ */
/* Matrix Market: io/matrix_market.hxx:99-254, io/detail/mmio_impl.hxx       */
/* ------------------------------------------------------------------------ */

enum {
  ORC_OK = 0,
  ORC_ERR_OPEN = 1,      /* matrix_market.hxx:111-114 */
  ORC_ERR_BANNER = 2,    /* :116-119 */
  ORC_ERR_ARRAY = 3,     /* :122-125 dense arrays rejected */
  ORC_ERR_SIZE = 4,      /* :129-133 */
  ORC_ERR_OVERFLOW = 5,  /* :135-140 */
  ORC_ERR_ENTRY = 6,     /* :161-166, :187-192 */
  ORC_ERR_TYPE = 7,      /* :198-201 complex etc. */
  ORC_ERR_ALLOC = 8
};

static void lower(char* s) {
  for (; *s; ++s) *s = (char)tolower((unsigned char)*s);
}

int orc_mtx_load(const char* path, orc_coo_t* out) {
  memset(out, 0, sizeof(*out));
  FILE* f = fopen(path, "r");
  if (!f) return ORC_ERR_OPEN;

  /* Banner: "%%MatrixMarket matrix <coordinate|array> <field> <symmetry>",
   * tokens compared case-insensitively (mmio_impl.hxx mm_read_banner). */
  char line[1025];
  char banner[64], mtx[64], crd[64], field[64], sym[64];
  if (!fgets(line, sizeof line, f)) { fclose(f); return ORC_ERR_BANNER; }
  if (sscanf(line, "%63s %63s %63s %63s %63s", banner, mtx, crd, field, sym) != 5) {
    fclose(f); return ORC_ERR_BANNER;
  }
  lower(mtx); lower(crd); lower(field); lower(sym);
  if (strncmp(banner, "%%MatrixMarket", 14) != 0 || strcmp(mtx, "matrix") != 0) {
    fclose(f); return ORC_ERR_BANNER;
  }
  int is_coordinate = strcmp(crd, "coordinate") == 0;
  int is_array = strcmp(crd, "array") == 0;
  if (!is_coordinate && !is_array) { fclose(f); return ORC_ERR_BANNER; }
  int is_real = strcmp(field, "real") == 0;
  int is_int = strcmp(field, "integer") == 0;
  int is_pattern = strcmp(field, "pattern") == 0;
  int is_complex = strcmp(field, "complex") == 0;
  if (!is_real && !is_int && !is_pattern && !is_complex) { fclose(f); return ORC_ERR_BANNER; }
  int is_general = strcmp(sym, "general") == 0;
  int is_symmetric = strcmp(sym, "symmetric") == 0;
  int is_herm = strcmp(sym, "hermitian") == 0;
  int is_skew = strcmp(sym, "skew-symmetric") == 0;
  if (!is_general && !is_symmetric && !is_herm && !is_skew) { fclose(f); return ORC_ERR_BANNER; }
  if (is_array) { fclose(f); return ORC_ERR_ARRAY; }

  /* Size line: skip comment lines, then "M N NNZ" (mm_read_mtx_crd_size). */
  size_t M = 0, N = 0, NZ = 0;
  for (;;) {
    if (!fgets(line, sizeof line, f)) { fclose(f); return ORC_ERR_SIZE; }
    if (line[0] != '%') break;
  }
  if (sscanf(line, "%zu %zu %zu", &M, &N, &NZ) != 3) {
    int r;
    do {
      r = fscanf(f, "%zu %zu %zu", &M, &N, &NZ);
      if (r == EOF) { fclose(f); return ORC_ERR_SIZE; }
    } while (r != 3);
  }
  if (M >= (size_t)INT_MAX || N >= (size_t)INT_MAX || NZ >= (size_t)INT_MAX) {
    fclose(f); return ORC_ERR_OVERFLOW;
  }
  if (!(is_pattern || is_real || is_int)) { fclose(f); return ORC_ERR_TYPE; }

  int32_t nnz = (int32_t)NZ;
  int32_t* I = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  int32_t* J = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  float* V = (float*)malloc(sizeof(float) * (size_t)(nnz > 0 ? nnz : 1));
  if (!I || !J || !V) { fclose(f); free(I); free(J); free(V); return ORC_ERR_ALLOC; }

  for (int32_t i = 0; i < nnz; ++i) {
    size_t r = 0, c = 0;
    double w = 1.0; /* pattern => weight 1.0, matrix_market.hxx:170-171 */
    int got;
    if (is_pattern) {
      got = fscanf(f, " %zu %zu \n", &r, &c);
      if (got != 2) goto bad_entry;
    } else {
      got = fscanf(f, " %zu %zu %lf \n", &r, &c, &w);
      if (got != 3) goto bad_entry;
    }
    if (r == 0 || c == 0) goto bad_entry; /* 1-based on disk */
    I[i] = (int32_t)r - 1;
    J[i] = (int32_t)c - 1;
    V[i] = (float)w; /* read as double, stored as weight_t: :196 */
    continue;
  bad_entry:
    fclose(f); free(I); free(J); free(V);
    return ORC_ERR_ENTRY;
  }
  fclose(f);

  out->rows = (int32_t)M;
  out->cols = (int32_t)N;
  out->weighted = is_pattern ? 0 : 1; /* :154, :174 */

  if (is_symmetric) {
    /* :203-246 every off-diagonal entry becomes the adjacent pair
     * (i,j),(j,i); diagonal entries stay single. */
    int32_t off = 0;
    for (int32_t i = 0; i < nnz; ++i)
      if (I[i] != J[i]) ++off;
    int32_t nn = 2 * off + (nnz - off);
    int32_t* I2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nn > 0 ? nn : 1));
    int32_t* J2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nn > 0 ? nn : 1));
    float* V2 = (float*)malloc(sizeof(float) * (size_t)(nn > 0 ? nn : 1));
    if (!I2 || !J2 || !V2) { free(I); free(J); free(V); free(I2); free(J2); free(V2); return ORC_ERR_ALLOC; }
    int32_t p = 0;
    for (int32_t i = 0; i < nnz; ++i) {
      I2[p] = I[i]; J2[p] = J[i]; V2[p] = V[i]; ++p;
      if (I[i] != J[i]) { I2[p] = J[i]; J2[p] = I[i]; V2[p] = V[i]; ++p; }
    }
    free(I); free(J); free(V);
    I = I2; J = J2; V = V2; nnz = nn;
    out->symmetric = 1; out->directed = 0; /* :204-205 */
  } else {
    out->symmetric = 0; out->directed = 1; /* :248-249 */
  }
  out->nnz = nnz;
  out->row_indices = I;
  out->column_indices = J;
  out->nonzero_values = V;
  return ORC_OK;
}

void orc_coo_free(orc_coo_t* c) {
  if (!c) return;
  free(c->row_indices); free(c->column_indices); free(c->nonzero_values);
  memset(c, 0, sizeof(*c));
}

/* ------------------------------------------------------------------------ */
/* COO -> CSR: formats/csr.hxx:81-140                                        */
/* ------------------------------------------------------------------------ */

int orc_csr_from_coo(const orc_coo_t* coo, orc_csr_t* out) {
  memset(out, 0, sizeof(*out));
  int32_t R = coo->rows, nnz = coo->nnz;
  int32_t* Ap = (int32_t*)calloc((size_t)R + 1, sizeof(int32_t));
  int32_t* Aj = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  float* Ax = (float*)malloc(sizeof(float) * (size_t)(nnz > 0 ? nnz : 1));
  if (!Ap || !Aj || !Ax) { free(Ap); free(Aj); free(Ax); return ORC_ERR_ALLOC; }

  for (int32_t n = 0; n < nnz; ++n) ++Ap[coo->row_indices[n]]; /* :105-107 */
  for (int32_t i = 0, sum = 0; i < R; ++i) {                   /* :110-114 */
    int32_t t = Ap[i]; Ap[i] = sum; sum += t;
  }
  Ap[R] = nnz;
  for (int32_t n = 0; n < nnz; ++n) {                          /* :119-127 */
    int32_t row = coo->row_indices[n];
    int32_t dest = Ap[row];
    Aj[dest] = coo->column_indices[n];
    Ax[dest] = coo->nonzero_values[n];
    ++Ap[row];
  }
  for (int32_t i = 0, last = 0; i <= R; ++i) {                 /* :129-133 */
    int32_t t = Ap[i]; Ap[i] = last; last = t;
  }
  out->rows = R; out->cols = coo->cols; out->nnz = nnz;
  out->row_offsets = Ap; out->column_indices = Aj; out->nonzero_values = Ax;
  return ORC_OK;
}

void orc_csr_free(orc_csr_t* c) {
  if (!c) return;
  free(c->row_offsets); free(c->column_indices); free(c->nonzero_values);
  memset(c, 0, sizeof(*c));
}

/* ------------------------------------------------------------------------ */
/* Binary min-heaps keyed on the tentative distance (std::priority_queue with */
/* the "p1.second > p2.second" comparator: bfs_cpu.hxx:13-19).                */
/* Tie order differs from libstdc++'s heap; the fixed point does not depend   */
/* on it.                                                                     */
/* ------------------------------------------------------------------------ */

typedef struct { int32_t v; int32_t d; } ent_i;
typedef struct { ent_i* a; size_t n, cap; } heap_i;

static int heap_i_push(heap_i* h, int32_t v, int32_t d) {
  if (h->n == h->cap) {
    size_t nc = h->cap ? h->cap * 2 : 1024;
    ent_i* na = (ent_i*)realloc(h->a, nc * sizeof(ent_i));
    if (!na) return -1;
    h->a = na; h->cap = nc;
  }
  size_t i = h->n++;
  while (i > 0) {
    size_t p = (i - 1) >> 1;
    if (h->a[p].d <= d) break;
    h->a[i] = h->a[p]; i = p;
  }
  h->a[i].v = v; h->a[i].d = d;
  return 0;
}

static ent_i heap_i_pop(heap_i* h) {
  ent_i top = h->a[0];
  ent_i last = h->a[--h->n];
  size_t i = 0, n = h->n;
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m;
    if (l >= n) break;
    m = (r < n && h->a[r].d < h->a[l].d) ? r : l;
    if (h->a[m].d >= last.d) break;
    h->a[i] = h->a[m]; i = m;
  }
  if (n) h->a[i] = last;
  return top;
}

double orc_bfs(int32_t nv, const int32_t* ro, const int32_t* ci, int32_t src,
               int32_t* dist) {
  for (int32_t i = 0; i < nv; ++i) dist[i] = INT32_MAX; /* bfs_cpu.hxx:32-33 */
  double t0 = now_ms();                                  /* :35 */
  dist[src] = 0;                                         /* :37 */
  heap_i pq = {0, 0, 0};
  heap_i_push(&pq, src, 0);                              /* :43 */
  while (pq.n) {                                         /* :45-63 */
    ent_i cur = heap_i_pop(&pq);
    int32_t u = cur.v, du = cur.d;
    for (int32_t e = ro[u]; e < ro[u + 1]; ++e) {
      int32_t nb = ci[e];
      int32_t nd = du + 1;
      if (nd < dist[nb]) {
        dist[nb] = nd;
        heap_i_push(&pq, nb, nd);
      }
    }
  }
  double t1 = now_ms();
  free(pq.a);
  return t1 - t0;
}

typedef struct { int32_t v; float d; } ent_f;
typedef struct { ent_f* a; size_t n, cap; } heap_f;

static int heap_f_push(heap_f* h, int32_t v, float d) {
  if (h->n == h->cap) {
    size_t nc = h->cap ? h->cap * 2 : 1024;
    ent_f* na = (ent_f*)realloc(h->a, nc * sizeof(ent_f));
    if (!na) return -1;
    h->a = na; h->cap = nc;
  }
  size_t i = h->n++;
  while (i > 0) {
    size_t p = (i - 1) >> 1;
    if (h->a[p].d <= d) break;
    h->a[i] = h->a[p]; i = p;
  }
  h->a[i].v = v; h->a[i].d = d;
  return 0;
}

static ent_f heap_f_pop(heap_f* h) {
  ent_f top = h->a[0];
  ent_f last = h->a[--h->n];
  size_t i = 0, n = h->n;
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m;
    if (l >= n) break;
    m = (r < n && h->a[r].d < h->a[l].d) ? r : l;
    if (h->a[m].d >= last.d) break;
    h->a[i] = h->a[m]; i = m;
  }
  if (n) h->a[i] = last;
  return top;
}

double orc_sssp(int32_t nv, const int32_t* ro, const int32_t* ci,
                const float* w, int32_t src, float* dist) {
  for (int32_t i = 0; i < nv; ++i) dist[i] = FLT_MAX; /* sssp_cpu.hxx:36-37 */
  double t0 = now_ms();
  dist[src] = 0.0f;                                    /* :41 */
  heap_f pq = {0, 0, 0};
  heap_f_push(&pq, src, 0.0f);                         /* :47 */
  while (pq.n) {                                       /* :49-67 */
    ent_f cur = heap_f_pop(&pq);
    int32_t u = cur.v;
    float du = cur.d;
    for (int32_t e = ro[u]; e < ro[u + 1]; ++e) {
      int32_t nb = ci[e];
      /* volatile keeps the add in fp32 storage precision on every target */
      volatile float nd = du + w[e];
      if (nd < dist[nb]) {
        dist[nb] = nd;
        heap_f_push(&pq, nb, nd);
      }
    }
  }
  double t1 = now_ms();
  free(pq.a);
  return t1 - t0;
}

/* orc_sssp with a time budget, for bench.py's bounded CPU sample: the same loop (sssp_cpu.hxx:49-67),
 * left after the pop that finds budget_ms exceeded (checked every 4096 pops).  *edges_scanned counts
 * the edges the loop looked at; *finished = 0 means the distances are NOT final. */
double orc_sssp_budget(int32_t nv, const int32_t* ro, const int32_t* ci,
                       const float* w, int32_t src, float* dist, double budget_ms,
                       int64_t* edges_scanned, int32_t* finished) {
  for (int32_t i = 0; i < nv; ++i) dist[i] = FLT_MAX;
  double t0 = now_ms();
  dist[src] = 0.0f;
  heap_f pq = {0, 0, 0};
  heap_f_push(&pq, src, 0.0f);
  int64_t scanned = 0, pops = 0;
  int32_t fin = 1;
  while (pq.n) {
    if (budget_ms > 0 && (++pops & 4095) == 0 && now_ms() - t0 > budget_ms) { fin = 0; break; }
    ent_f cur = heap_f_pop(&pq);
    int32_t u = cur.v;
    float du = cur.d;
    scanned += ro[u + 1] - ro[u];
    for (int32_t e = ro[u]; e < ro[u + 1]; ++e) {
      int32_t nb = ci[e];
      volatile float nd = du + w[e];
      if (nd < dist[nb]) {
        dist[nb] = nd;
        heap_f_push(&pq, nb, nd);
      }
    }
  }
  double t1 = now_ms();
  free(pq.a);
  if (edges_scanned) *edges_scanned = scanned;
  if (finished) *finished = fin;
  return t1 - t0;
}

double orc_bfs_queue(int32_t nv, const int32_t* ro, const int32_t* ci,
                     int32_t src, int32_t* dist, int64_t* edges_visited) {
  for (int32_t i = 0; i < nv; ++i) dist[i] = INT32_MAX;
  int32_t* q = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nv > 0 ? nv : 1));
  double t0 = now_ms();
  size_t head = 0, tail = 0;
  int64_t ev = 0;
  dist[src] = 0; q[tail++] = src;
  while (head < tail) {
    int32_t u = q[head++];
    int32_t nd = dist[u] + 1;
    ev += ro[u + 1] - ro[u];
    for (int32_t e = ro[u]; e < ro[u + 1]; ++e) {
      int32_t nb = ci[e];
      if (dist[nb] == INT32_MAX) { dist[nb] = nd; q[tail++] = nb; }
    }
  }
  double t1 = now_ms();
  free(q);
  if (edges_visited) *edges_visited = ev;
  return t1 - t0;
}

/* ------------------------------------------------------------------------ */
/* PageRank: algorithms/pr.hxx                                               */
/* ------------------------------------------------------------------------ */

int orc_pr_f32(int32_t nv, const int32_t* ro, const int32_t* ci, const float* w,
               float alpha, float tol, int max_iter, float* p, double* ms) {
  float* plast = (float*)malloc(sizeof(float) * (size_t)(nv > 0 ? nv : 1));
  float* iw = (float*)malloc(sizeof(float) * (size_t)(nv > 0 ? nv : 1));
  /* reset(), pr.hxx:65-93 */
  float p0 = (float)(1.0 / (double)nv); /* fill_n(p, n, 1.0 / n_vertices) :74 */
  for (int32_t i = 0; i < nv; ++i) { p[i] = p0; plast[i] = 0.0f; }
  for (int32_t i = 0; i < nv; ++i) { /* :78-88 */
    volatile float val = 0.0f;
    for (int32_t e = ro[i]; e < ro[i + 1]; ++e) val = val + w[e];
    iw[i] = (val != 0.0f) ? alpha / val : 0.0f;
  }
  double t0 = now_ms();
  int iteration = 0;
  for (;;) {
    /* is_converged(), :172-195; checked before every loop() (enactor.hxx:274) */
    if (iteration > 0) {
      float err = 0.0f;
      for (int32_t i = 0; i < nv; ++i) {
        float d = fabsf(p[i] - plast[i]);
        if (d > err) err = d;
      }
      if (err < tol) break;
    }
    if (max_iter > 0 && iteration >= max_iter) break;
    /* loop(), :107-152 */
    memcpy(plast, p, sizeof(float) * (size_t)nv); /* :121 */
    volatile float dsum = 0.0f;                   /* :125-132 */
    for (int32_t i = 0; i < nv; ++i)
      dsum = dsum + (iw[i] == 0.0f ? alpha * p[i] : 0.0f);
    volatile float num = (1 - alpha) + dsum;      /* (1 - alpha + dsum) :134 */
    float base = num / (float)nv;
    for (int32_t i = 0; i < nv; ++i) p[i] = base;
    for (int32_t s = 0; s < nv; ++s) {            /* edge-parallel op :140-146 */
      for (int32_t e = ro[s]; e < ro[s + 1]; ++e) {
        volatile float t = plast[s] * iw[s];
        volatile float upd = t * w[e];
        volatile float acc = p[ci[e]] + upd;
        p[ci[e]] = acc;
      }
    }
    ++iteration;
  }
  double t1 = now_ms();
  if (ms) *ms = t1 - t0;
  free(plast); free(iw);
  return iteration;
}

int orc_pr_f64(int32_t nv, const int32_t* ro, const int32_t* ci, const float* w,
               double alpha, double tol, int max_iter, int force_iter,
               double* p, double* ms) {
  double* plast = (double*)malloc(sizeof(double) * (size_t)(nv > 0 ? nv : 1));
  double* iw = (double*)malloc(sizeof(double) * (size_t)(nv > 0 ? nv : 1));
  for (int32_t i = 0; i < nv; ++i) { p[i] = 1.0 / (double)nv; plast[i] = 0.0; }
  for (int32_t i = 0; i < nv; ++i) {
    double val = 0.0;
    for (int32_t e = ro[i]; e < ro[i + 1]; ++e) val += (double)w[e];
    iw[i] = (val != 0.0) ? alpha / val : 0.0;
  }
  double t0 = now_ms();
  int iteration = 0;
  for (;;) {
    if (force_iter > 0) {
      if (iteration >= force_iter) break;
    } else {
      if (iteration > 0) {
        double err = 0.0;
        for (int32_t i = 0; i < nv; ++i) {
          double d = fabs(p[i] - plast[i]);
          if (d > err) err = d;
        }
        if (err < tol) break;
      }
      if (max_iter > 0 && iteration >= max_iter) break;
    }
    memcpy(plast, p, sizeof(double) * (size_t)nv);
    double dsum = 0.0;
    for (int32_t i = 0; i < nv; ++i) dsum += (iw[i] == 0.0 ? alpha * p[i] : 0.0);
    double base = (1.0 - alpha + dsum) / (double)nv;
    for (int32_t i = 0; i < nv; ++i) p[i] = base;
    for (int32_t s = 0; s < nv; ++s) {
      double x = plast[s] * iw[s];
      for (int32_t e = ro[s]; e < ro[s + 1]; ++e) p[ci[e]] += x * (double)w[e];
    }
    ++iteration;
  }
  double t1 = now_ms();
  if (ms) *ms = t1 - t0;
  free(plast); free(iw);
  return iteration;
}

/* ------------------------------------------------------------------------ */
/* Fixed-point property checkers (size independent)                          */
/* ------------------------------------------------------------------------ */

int64_t orc_check_bfs(int32_t nv, const int32_t* ro, const int32_t* ci,
                      int32_t src, const int32_t* d) {
  /* d is THE BFS depth vector iff: d[src]=0; for every edge (u,v) with u
   * reached, d[v] <= d[u]+1; every reached v != src has an in-edge from some
   * u with d[u] = d[v]-1; nothing else is 0. Checked with one pass that
   * marks "has tight parent". */
  int64_t bad = 0;
  unsigned char* tight = (unsigned char*)calloc((size_t)(nv > 0 ? nv : 1), 1);
  if (d[src] != 0) ++bad;
  for (int32_t u = 0; u < nv; ++u) {
    if (d[u] == INT32_MAX) continue;
    if (d[u] < 0) { ++bad; continue; }
    for (int32_t e = ro[u]; e < ro[u + 1]; ++e) {
      int32_t v = ci[e];
      if (d[v] > d[u] + 1) ++bad;      /* includes unreached v */
      if (d[v] == d[u] + 1) tight[v] = 1;
    }
  }
  for (int32_t v = 0; v < nv; ++v) {
    if (v == src || d[v] == INT32_MAX) continue;
    if (d[v] == 0 || !tight[v]) ++bad;
  }
  free(tight);
  return bad;
}

int64_t orc_check_sssp(int32_t nv, const int32_t* ro, const int32_t* ci,
                       const float* w, int32_t src, const float* d) {
  /* Fixed point of d[v] = min_u fl(d[u] + w(u,v)), d[src] = 0, unreached =
   * FLT_MAX: no edge can still relax, and every reached v != src is tight on
   * some in-edge.  For positive weights that is the unique Dijkstra result. */
  int64_t bad = 0;
  unsigned char* tight = (unsigned char*)calloc((size_t)(nv > 0 ? nv : 1), 1);
  if (d[src] != 0.0f) ++bad;
  for (int32_t u = 0; u < nv; ++u) {
    if (d[u] == FLT_MAX) continue;
    for (int32_t e = ro[u]; e < ro[u + 1]; ++e) {
      int32_t v = ci[e];
      volatile float nd = d[u] + w[e];
      if (nd < d[v]) ++bad;
      if (nd == d[v]) tight[v] = 1;
    }
  }
  for (int32_t v = 0; v < nv; ++v) {
    if (v == src || d[v] == FLT_MAX) continue;
    if (!tight[v]) ++bad;
  }
  free(tight);
  return bad;
}
