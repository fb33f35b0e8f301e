/* CPU design study (not product code): how many relaxations does the level-synchronous schedule of the binned relaxation
 * (grx_relax.hpp: every relaxation of a level reads the labels as they were when the level BEGAN -- Jacobi) spend, and how
 * many would it spend if a level were run as K sequential SUB-LEVELS (the frontier, which is in ascending vertex order, cut
 * into K contiguous parts by out-edges; part j reads the labels parts < j have already lowered)?  The fixed point is the
 * same for every K; what changes is the work (edges relaxed) and the number of scatter + sweep pairs.
 *   distances are checked against a binary-heap Dijkstra.
 *   built and driven by tools/proto/jacobi_sublevels.py (ctypes).                                                     */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float d; int v; } item_t;

/* out[0] = levels, out[1] = sub-level passes, out[2] = edges relaxed, out[3] = vertices relaxed, out[4] = mismatches against
 * Dijkstra; per_level (may be NULL): [level] = edges relaxed, up to cap entries.  mode 0: parts by out-edges in vertex order;
 * mode 1: parts by ascending label (the nearest 1/K of the frontier's out-edges first: a near-far split of ONE level) */
int jacobi_sublevels(int V, const int* ro, const int* ci, const float* w, int src, int K, int mode, long long* out,
                     long long* per_level, int cap) {
  float* d = malloc(sizeof(float) * V);
  float* dstart = malloc(sizeof(float) * V);  /* label of a frontier vertex when its PART begins */
  int* stamp = malloc(sizeof(int) * V);       /* level in which the vertex entered the next frontier */
  int* f0 = malloc(sizeof(int) * V);
  int* f1 = malloc(sizeof(int) * V);
  int* order = malloc(sizeof(int) * V);
  for (int i = 0; i < V; ++i) { d[i] = FLT_MAX; stamp[i] = -1; }
  d[src] = 0.0f;
  int n = 1, level = 0;
  f0[0] = src;
  long long edges = 0, verts = 0, passes = 0;
  while (n > 0) {
    /* frontier in ascending vertex order (what the sweep of the previous level emits) */
    /* f0 is built in ascending order below (a counting pass over stamps would do the same); sort defensively */
    int sorted = 1;
    for (int i = 1; i < n; ++i) if (f0[i - 1] > f0[i]) { sorted = 0; break; }
    if (!sorted) {
      /* radix-free: mark + sweep */
      char* mark = calloc(V, 1);
      for (int i = 0; i < n; ++i) mark[f0[i]] = 1;
      int k = 0;
      for (int v = 0; v < V; ++v) if (mark[v]) f0[k++] = v;
      free(mark);
    }
    long long m = 0;
    for (int i = 0; i < n; ++i) m += ro[f0[i] + 1] - ro[f0[i]];
    if (per_level && level < cap) per_level[level] = m;
    edges += m;
    verts += n;
    int parts = K;
    if (m < (1 << 20)) parts = 1;  /* thin levels are not binned: one pass */
    for (int i = 0; i < n; ++i) order[i] = f0[i];
    if (mode == 1 && parts > 1) {
      /* ascending label: insertion into buckets would do on the device; here a plain sort by (label, id) */
      /* simple shell sort on labels (n up to millions: use qsort via global) */
      /* -- qsort with a static pointer */
      static float* g_d;
      g_d = d;
      int cmp(const void* a, const void* b) {
        const float x = g_d[*(const int*)a], y = g_d[*(const int*)b];
        return x < y ? -1 : (x > y ? 1 : (*(const int*)a - *(const int*)b));
      }
      qsort(order, n, sizeof(int), cmp);
    }
    int n1 = 0;
    long long per = (m + parts - 1) / parts, acc = 0;
    int i0 = 0;
    for (int p = 0; p < parts && i0 < n; ++p) {
      int i1 = i0;
      long long em = 0;
      while (i1 < n && (em < per || p == parts - 1)) { em += ro[order[i1] + 1] - ro[order[i1]]; ++i1; }
      acc += em;
      ++passes;
      /* scatter: candidates from the labels as they are NOW (start of this part); sweep: minima applied afterwards.
       * Jacobi inside the part: read dstart, write d only after all candidates of the part are formed -- emulate with a
       * snapshot of the sources' labels (targets may be sources of the same part: their label must not move under them) */
      for (int i = i0; i < i1; ++i) dstart[order[i]] = d[order[i]];
      for (int i = i0; i < i1; ++i) {
        const int v = order[i];
        const float dv = dstart[v];
        for (int e = ro[v]; e < ro[v + 1]; ++e) {
          const float nd = dv + w[e];
          const int u = ci[e];
          if (nd < d[u]) {
            /* (min over all candidates of the part: applying them one by one gives the same minimum) */
            d[u] = nd;
            if (stamp[u] != level) { stamp[u] = level; f1[n1++] = u; }
          }
        }
      }
      /* NOTE: lowering d[u] inside the loop is visible to a LATER source of the same part only through dstart, which was
       * snapshotted above -- so the part is Jacobi, the parts among themselves Gauss-Seidel */
      i0 = i1;
    }
    /* next frontier ascending */
    {
      char* mark = calloc(V, 1);
      for (int i = 0; i < n1; ++i) mark[f1[i]] = 1;
      int k = 0;
      for (int v = 0; v < V; ++v) if (mark[v]) f0[k++] = v;
      free(mark);
      n = k;
    }
    ++level;
  }
  /* Dijkstra */
  long long bad = 0;
  {
    float* dd = malloc(sizeof(float) * V);
    for (int i = 0; i < V; ++i) dd[i] = FLT_MAX;
    size_t capn = (size_t)ro[V] + 16;
    item_t* heap = malloc(sizeof(item_t) * capn);
    size_t hn = 0;
    dd[src] = 0.0f;
    heap[hn].d = 0.0f; heap[hn].v = src; ++hn;
    while (hn > 0) {
      item_t top = heap[0];
      item_t last = heap[--hn];
      size_t i = 0;
      for (;;) { size_t c = 2 * i + 1; if (c >= hn) break; if (c + 1 < hn && heap[c + 1].d < heap[c].d) ++c; if (heap[c].d >= last.d) break; heap[i] = heap[c]; i = c; }
      if (hn > 0) heap[i] = last;
      if (top.d > dd[top.v]) continue;
      for (int e = ro[top.v]; e < ro[top.v + 1]; ++e) {
        const float nd = top.d + w[e];
        const int u = ci[e];
        if (nd < dd[u]) {
          dd[u] = nd;
          size_t j = hn++;
          while (j > 0) { size_t p = (j - 1) / 2; if (heap[p].d <= nd) break; heap[j] = heap[p]; j = p; }
          heap[j].d = nd; heap[j].v = u;
        }
      }
    }
    for (int i = 0; i < V; ++i) if (dd[i] != d[i]) ++bad;
    free(heap);
    free(dd);
  }
  out[0] = level; out[1] = passes; out[2] = edges; out[3] = verts; out[4] = bad;
  free(d); free(dstart); free(stamp); free(f0); free(f1); free(order);
  return 0;
}
