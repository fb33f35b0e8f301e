// grx_transpose.hip -- one-time device build of the transpose (CSC) of a CSR
// graph: the layout behind the pull operators (PageRank pull, bottom-up BFS).
// The reference has CSC as a separate user-built view (include/gunrock/graph/csc.hxx,
// formats/csc.hxx) and its advance ignores `direction` (operators/configs.hxx:78-82);
// here the engine derives it lazily and caches it in the graph handle.
//
// Column order: positions inside a column are claimed with an atomic cursor, so
// the order of a column's entries is not reproducible run to run (fp32 sums over
// in-edges may differ in the last ulp, as the reference's atomicAdd order does).
#include "grx_engine.hpp"
#include <gunrock/hip/scan.hxx>

#include <algorithm>
#include <climits>
#include <cstdlib>

namespace grx {

__global__ void tr_count_kernel(const int32_t* __restrict__ ci, int64_t E, int32_t* cnt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride)
    atomicAdd(&cnt[ci[e]], 1);
}

// One wave per CSR row chunk: rows are walked in order by a grid-stride loop
// over rows; each lane takes edges of the row.  Position inside the column is
// claimed with an atomic cursor, so a per-column sort follows.
__global__ void tr_fill_kernel(const int32_t* __restrict__ ro, const int32_t* __restrict__ ci,
                               const float* __restrict__ w, int32_t V, int32_t* cursor,
                               int32_t* t_ci, float* t_w, int32_t deg_lo, int32_t deg_hi) {
  const int lane = dev::lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t v = wave; v < V; v += nwaves) {
    const int b = ro[v], e = ro[v + 1];
    if (e - b < deg_lo || e - b >= deg_hi) continue;  // this pass places sources of another degree class
    for (int k = b + lane; k < e; k += 64) {
      const int dst = ci[k];
      const int pos = atomicAdd(&cursor[dst], 1);
      t_ci[pos] = (int32_t)v;
      if (t_w) t_w[pos] = w ? w[k] : 1.0f;
    }
  }
}

// ---- is the CSR its own transpose? ---------------------------------------------------------
// graph_properties_t::symmetric is caller-supplied, defaults to true and is inert in the
// reference (graph/properties.hxx:13-18); here it is LOAD-BEARING: a symmetric graph's CSR
// doubles as its in-edge list in the bottom-up step.  So the claim is verified once per graph
// handle: for every vertex the multiset of out-neighbours must equal the multiset of
// in-neighbours, compared through 64-bit sums of a mixing hash of the neighbour ids (one pass
// over the edges; a wrong "equal" needs a 2^-64 collision).
__device__ __forceinline__ unsigned long long sym_mix(unsigned long long x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
__global__ void sym_hash_kernel(const int32_t* __restrict__ ro, const int32_t* __restrict__ ci, int32_t V,
                                unsigned long long* h_in, unsigned long long* h_out, int32_t* bad) {
  const int lane = dev::lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t u = wave; u < V; u += nwaves) {
    const int b = ro[u], e = ro[u + 1];
    const unsigned long long hu = sym_mix((unsigned long long)u);
    unsigned long long acc = 0ull;
    for (int k = b + lane; k < e; k += 64) {
      const int v = ci[k];
      if (v < 0 || v >= V) { *bad = 1; continue; }
      acc += sym_mix((unsigned long long)v);
      atomicAdd(&h_in[v], hu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) h_out[u] = acc;
  }
}
__global__ void sym_compare_kernel(const unsigned long long* h_in, const unsigned long long* h_out, int32_t V,
                                   int32_t* bad) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x)
    if (h_in[v] != h_out[v]) *bad = 1;
}

}  // namespace grx

using namespace grx;

grx_status_t grx::graph_is_symmetric(grx_context_t ctx, grx_graph_t g, bool* result) {
  if (g->sym_checked == 0) {
    const int32_t V = g->V;
    hipStream_t s = ctx->stream;
    unsigned long long* h = nullptr;
    int32_t* bad = nullptr;
    GRX_HIP(hipMalloc(reinterpret_cast<void**>(&h), (2 * (size_t)V + 2) * sizeof(unsigned long long)));
    GRX_HIP(hipMalloc(reinterpret_cast<void**>(&bad), sizeof(int32_t)));
    GRX_HIP(hipMemsetAsync(h, 0, (2 * (size_t)V + 2) * sizeof(unsigned long long), s));
    GRX_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(sym_hash_kernel, dim3(2048), dim3(256), 0, s, g->ro, g->ci, V, h, h + V, bad);
    hipLaunchKernelGGL(sym_compare_kernel, dim3(1024), dim3(256), 0, s, h, h + V, V, bad);
    int32_t hb = 0;
    GRX_HIP(hipMemcpyAsync(&hb, bad, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    GRX_HIP(hipStreamSynchronize(s));
    (void)hipFree(h);
    (void)hipFree(bad);
    GRX_HIP(hipGetLastError());
    g->sym_checked = hb ? 2 : 1;
  }
  *result = g->sym_checked == 1;
  return GRX_SUCCESS;
}

grx_status_t grx::graph_build_transpose(grx_context_t ctx, grx_graph_t g) {
  if (g->has_transpose) return GRX_SUCCESS;
  const int32_t V = g->V;
  const int64_t E = g->E;
  hipStream_t s = ctx->stream;
  GRX_HIP(hipMalloc(reinterpret_cast<void**>(&g->t_ro), ((size_t)V + 2) * sizeof(int32_t)));
  GRX_HIP(hipMalloc(reinterpret_cast<void**>(&g->t_ci), (size_t)(E > 0 ? E : 1) * sizeof(int32_t)));
  const bool weighted = g->w != nullptr;
  if (weighted) GRX_HIP(hipMalloc(reinterpret_cast<void**>(&g->t_w), (size_t)(E > 0 ? E : 1) * sizeof(float)));
  int32_t *cnt = nullptr, *bs = nullptr;
  GRX_HIP(hipMalloc(reinterpret_cast<void**>(&cnt), ((size_t)V + 2) * sizeof(int32_t)));
  GRX_HIP(hipMalloc(reinterpret_cast<void**>(&bs), ((size_t)scan_num_blocks(V + 1) + 2) * sizeof(int32_t)));
  GRX_HIP(hipMemsetAsync(cnt, 0, ((size_t)V + 2) * sizeof(int32_t), s));
  if (E > 0) hipLaunchKernelGGL(tr_count_kernel, dim3(2048), dim3(256), 0, s, g->ci, E, cnt);
  exclusive_scan_i32(s, cnt, (int64_t)V, g->t_ro, bs);
  GRX_HIP(hipMemcpyAsync(cnt, g->t_ro, ((size_t)V + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (E > 0) {
    // HUBS FIRST inside every in-list: three passes over the sources by out-degree class (>= 16x the
    // mean, >= the mean, the rest) share the column cursors, so a bottom-up BFS level -- which
    // stops at the first in-neighbour found in the frontier -- meets the likeliest parents first.
    const int32_t mean = (int32_t)std::max<int64_t>(1, E / std::max(1, V));
    const bool tiers = getenv("GRX_TR_NOTIERS") == nullptr;
    const int32_t bounds[4] = {INT32_MAX, tiers ? 16 * mean : 0, tiers ? mean : 0, 0};
    for (int t = 0; t < 3; ++t) {
      if (bounds[t] == bounds[t + 1]) continue;
      hipLaunchKernelGGL(tr_fill_kernel, dim3(2048), dim3(256), 0, s, g->ro, g->ci, g->w, V, cnt, g->t_ci, g->t_w,
                         bounds[t + 1], bounds[t]);
    }
  }
  GRX_HIP(hipStreamSynchronize(s));
  std::vector<int32_t> h_ro((size_t)V + 1);
  GRX_HIP(hipMemcpy(h_ro.data(), g->t_ro, ((size_t)V + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
  (void)hipFree(cnt);
  (void)hipFree(bs);
  GRX_HIP(hipGetLastError());
  g->h_t_ro.swap(h_ro);
  g->has_transpose = true;
  return GRX_SUCCESS;
}
