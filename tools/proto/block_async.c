/* CPU prototype of the block-asynchronous relaxation schedule for road-like graphs (round 4 design study; not product
 * code): partition by BFS region growing into blocks of <= NV vertices / <= NE intra edges, then synchronous supersteps in
 * which every active block relaxes to its LOCAL fixed point (label-correcting rounds) and pushes boundary updates to its
 * neighbours.  Reports supersteps, activations, rounds and work inflation against one relaxation per edge, and checks the
 * labels against a plain Dijkstra / BFS.
 *   gcc -O2 -o /tmp/block_async tools/proto/block_async.c -L gunrock_amd -lgrx -Wl,-rpath,$PWD/gunrock_amd -lm
 *   /tmp/block_async <side> <weighted 0|1> <NV> <NE> */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/grx.h"

typedef struct { float d; int v; } item_t;
static item_t* heap; static int hn;
static void hpush(float d, int v) { int i = hn++; while (i > 0) { int p = (i - 1) / 2; if (heap[p].d <= d) break; heap[i] = heap[p]; i = p; } heap[i].d = d; heap[i].v = v; }
static item_t hpop(void) { item_t top = heap[0]; item_t last = heap[--hn]; int i = 0; for (;;) { int c = 2 * i + 1; if (c >= hn) break; if (c + 1 < hn && heap[c + 1].d < heap[c].d) ++c; if (heap[c].d >= last.d) break; heap[i] = heap[c]; i = c; } heap[i] = last; return top; }

int main(int argc, char** argv) {
  int side = argc > 1 ? atoi(argv[1]) : 1000, weighted = argc > 2 ? atoi(argv[2]) : 0;
  int NV = argc > 3 ? atoi(argv[3]) : 8192, NE = argc > 4 ? atoi(argv[4]) : 24576;
  float DELTA = argc > 5 ? (float)atof(argv[5]) : 0.0f; /* 0: every active block every superstep */
  grx_host_csr_t h;
  if (grx_host_csr_generate(2, side * side, 0, 0.602f, 0.0f, weighted ? 1.0f : 0.0f, 42, &h) != GRX_SUCCESS) { printf("gen failed\n"); return 1; }
  int V, E, dir, wt, sym;
  grx_host_csr_info(h, &V, &E, &dir, &wt, &sym);
  const int32_t* ro = grx_host_csr_row_offsets(h); const int32_t* ci = grx_host_csr_column_indices(h); const float* w = grx_host_csr_values(h);
  int src = (side / 2) * side + side / 2;
  printf("V %d E %d src %d weighted %d NV %d NE %d\n", V, E, src, weighted, NV, NE);
  /* ---- partition: BFS region growing in id order */
  int* blk = malloc(sizeof(int) * V); int* loc = malloc(sizeof(int) * V); memset(blk, -1, sizeof(int) * V);
  int* queue = malloc(sizeof(int) * (NV + 8)); int nb = 0;
  int* bcount = malloc(sizeof(int) * (V / 16 + 16));
  {
    int qh = 0, qt = 0, ne = 0;
    for (int s = 0; s < V; ++s) {
      if (blk[s] >= 0) continue;
      /* a block is filled with whole BFS regions one after the other (small components are packed together) */
      if (qt >= NV || ne + 8 >= NE) { for (int i = 0; i < qt; ++i) loc[queue[i]] = i; bcount[nb++] = qt; qh = qt = ne = 0; }
      queue[qt++] = s; blk[s] = nb;
      while (qh < qt) {
        int u = queue[qh++];
        ne += ro[u + 1] - ro[u];
        for (int e = ro[u]; e < ro[u + 1]; ++e) { int v = ci[e]; if (blk[v] < 0 && qt < NV && ne + (qt - qh + 1) * 4 < NE) { blk[v] = nb; queue[qt++] = v; } }
      }
    }
    if (qt) { for (int i = 0; i < qt; ++i) loc[queue[i]] = i; bcount[nb++] = qt; }
  }
  long long cross = 0; int small = 0;
  for (int u = 0; u < V; ++u) for (int e = ro[u]; e < ro[u + 1]; ++e) if (blk[ci[e]] != blk[u]) ++cross;
  for (int b = 0; b < nb; ++b) if (bcount[b] < NV / 4) ++small;
  printf("blocks %d (mean %.0f vertices, %d below NV/4), cross edges %lld = %.2f%%\n", nb, (double)V / nb, small, cross, 100.0 * cross / E);
  /* block membership lists */
  int* bstart = malloc(sizeof(int) * (nb + 1)); bstart[0] = 0; for (int b = 0; b < nb; ++b) bstart[b + 1] = bstart[b] + bcount[b];
  int* members = malloc(sizeof(int) * V); { int* fill = calloc(nb, sizeof(int)); for (int v = 0; v < V; ++v) members[bstart[blk[v]] + fill[blk[v]]++] = v; free(fill); }
  /* ---- reference labels */
  float* ref = malloc(sizeof(float) * V); for (int i = 0; i < V; ++i) ref[i] = 3.4028235e38f;
  heap = malloc(sizeof(item_t) * ((size_t)E + 16)); hn = 0; ref[src] = 0; hpush(0, src);
  long long useful = 0;
  while (hn) { item_t t = hpop(); if (t.d > ref[t.v]) continue; useful += ro[t.v + 1] - ro[t.v]; for (int e = ro[t.v]; e < ro[t.v + 1]; ++e) { float nd = t.d + w[e]; if (nd < ref[ci[e]]) { ref[ci[e]] = nd; hpush(nd, ci[e]); } } }
  /* ---- block-async supersteps inside global buckets [lo, hi): a vertex is EXPANDED only while its label is < hi; labels
   * below hi are final when the bucket has no active block left (delta-stepping), the inner loops are block-local */
  float* dist = malloc(sizeof(float) * V); float* expd = malloc(sizeof(float) * V);
  for (int i = 0; i < V; ++i) dist[i] = expd[i] = 3.4028235e38f;
  dist[src] = 0;
  char* active = calloc(nb, 1); char* active_next = calloc(nb, 1); active[blk[src]] = 1;
  char* infr = calloc(V, 1); int* fr = malloc(sizeof(int) * V); int* fr2 = malloc(sizeof(int) * V);
  float* bmin = malloc(sizeof(float) * nb); long long sum_active = 0; /* bmin: smallest pending label of the block */
  for (int b = 0; b < nb; ++b) bmin[b] = 3.4028235e38f;
  bmin[blk[src]] = 0;
  long long relax = 0, activations = 0, rounds_total = 0, max_rounds = 0, supersteps = 0, max_active = 0, empty_act = 0, buckets = 0;
  float hi = DELTA > 0 ? DELTA : 3.4028235e38f;
  long long np = 0, pcap = 1 << 20; int* pv = malloc(sizeof(int) * pcap); float* pd = malloc(sizeof(float) * pcap);
  for (;;) {
    int n_active = 0;
    for (int b = 0; b < nb; ++b) { active[b] = bmin[b] < hi; n_active += active[b]; }
    if (!n_active) {
      float gmin = 3.4028235e38f; for (int b = 0; b < nb; ++b) if (bmin[b] < gmin) gmin = bmin[b];
      if (gmin >= 3e38f) break;
      hi = gmin + DELTA; ++buckets; continue;
    }
    ++supersteps; if (n_active > max_active) max_active = n_active; sum_active += n_active;
    for (int b = 0; b < nb; ++b) {
      if (!active[b]) continue;
      ++activations;
      int nf = 0;
      for (int i = bstart[b]; i < bstart[b + 1]; ++i) { int v = members[i]; if (dist[v] < expd[v] && dist[v] < hi) { fr[nf++] = v; } }
      if (!nf) ++empty_act;
      int rounds = 0;
      while (nf) {
        ++rounds; int nf2 = 0;
        for (int i = 0; i < nf; ++i) { int u = fr[i]; expd[u] = dist[u];
          for (int e = ro[u]; e < ro[u + 1]; ++e) { int v = ci[e]; ++relax; float nd = dist[u] + w[e];
            if (blk[v] != b) { if (nd < dist[v]) { if (np == pcap) { pcap *= 2; pv = realloc(pv, sizeof(int) * pcap); pd = realloc(pd, sizeof(float) * pcap); } pv[np] = v; pd[np++] = nd; } }  /* JACOBI: seen next superstep */
            else if (nd < dist[v]) { dist[v] = nd; if (nd < hi && !infr[v]) { infr[v] = 1; fr2[nf2++] = v; } } } }
        for (int i = 0; i < nf2; ++i) infr[fr2[i]] = 0;
        int* t = fr; fr = fr2; fr2 = t; nf = nf2;
      }
      rounds_total += rounds; if (rounds > max_rounds) max_rounds = rounds;
      float m = 3.4028235e38f; for (int i = bstart[b]; i < bstart[b + 1]; ++i) { int v = members[i]; if (dist[v] < expd[v] && dist[v] < m) m = dist[v]; }
      bmin[b] = m;
    }
    for (long long i = 0; i < np; ++i) { int v = pv[i]; if (pd[i] < dist[v]) { dist[v] = pd[i]; if (pd[i] < bmin[blk[v]]) bmin[blk[v]] = pd[i]; } }
    np = 0;
  }
  printf("buckets %lld\n", buckets);
  long long bad = 0; for (int i = 0; i < V; ++i) if (dist[i] != ref[i]) ++bad;
  printf("supersteps %lld, activations %lld (%lld empty), max active blocks %lld, rounds total %lld (max %lld, mean %.1f per activation)\n", supersteps, activations, empty_act, max_active, rounds_total, max_rounds, (double)rounds_total / activations);
  printf("delta %g: mean active blocks per superstep %.1f\n", DELTA, (double)sum_active / supersteps);
  printf("relaxations %lld vs useful %lld: inflation %.2fx ; mismatches vs Dijkstra %lld\n", relax, useful, (double)relax / useful, bad);
  return bad != 0;
}
