// grx_transpose.hip -- one-time device build of the transpose (CSC) of a CSR
// graph: the layout behind the pull operators (PageRank pull, bottom-up BFS).
// The reference has CSC as a separate user-built view (include/gunrock/graph/csc.hxx,
// formats/csc.hxx) and its advance ignores `direction` (operators/configs.hxx:78-82);
// here the engine derives it lazily and caches it in the graph handle.
//
// Column order (round 4): the edges are radix-sorted by (destination, hub tier of the source), stably -- a column lists
// its hub sources first and is otherwise in ascending source order, identically on every run and handle (round 1-3
// claimed positions with an atomic cursor per edge: 11 ms for 69 M edges and an order that changed from run to run).
#include "grx_engine.hpp"
#include "grx_sort.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>

namespace grx {

// (key, value[, weight]) of every edge, in CSR order: key = destination << 2 | tier of the SOURCE, value = source.
// Tier 0: out-degree >= 16 x the mean, 1: >= the mean, 2: the rest -- sorted by this key a column lists its hub sources
// first (a bottom-up BFS level stops at the first in-neighbour it finds in the frontier: the likeliest parents come first),
// and inside a tier in ascending source order.  (edge_expand_kernel, grx_sort.hpp, supplies the source row of every edge.)
struct tr_emit {
  const int32_t* ro;
  const float* w;
  int32_t deg_hub, deg_mean;
  uint32_t* keys;
  uint32_t* vals;
  uint32_t* vals2;
  __device__ __forceinline__ void operator()(int64_t e, int row, int col) const {
    const int deg = ro[row + 1] - ro[row];
    const uint32_t tier = deg >= deg_hub ? 0u : (deg >= deg_mean ? 1u : 2u);
    keys[e] = ((uint32_t)col << 2) | tier;
    vals[e] = (uint32_t)row;
    if (vals2) vals2[e] = w ? __float_as_uint(w[e]) : 0x3f800000u;
  }
};

// ---- is the CSR its own transpose? ---------------------------------------------------------
// graph_properties_t::symmetric is caller-supplied, defaults to true and is inert in the
// reference (graph/properties.hxx:13-18); here it is LOAD-BEARING: a symmetric graph's CSR
// doubles as its in-edge list in the bottom-up step.  So the claim is verified once per graph
// handle: for every vertex the multiset of out-neighbours must equal the multiset of
// in-neighbours, compared through 64-bit sums of a mixing hash of the neighbour ids (a wrong
// "equal" needs a 2^-64 collision).  Round 4: the in-neighbours are read from the TRANSPOSE
// (built by the stable sort, grx_sort.hpp) -- two row-local sums per vertex, no atomics; rounds
// 1-3 scattered one 64-bit atomicAdd per edge (49 ms for the 182 M edges of the kron stand-in).
__device__ __forceinline__ unsigned long long sym_mix(unsigned long long x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
__global__ void sym_compare_rows_kernel(const int32_t* __restrict__ ro, const int32_t* __restrict__ ci,
                                        const int32_t* __restrict__ t_ro, const int32_t* __restrict__ t_ci, int32_t V, int32_t* bad) {
  const int lane = dev::lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t u = wave; u < V; u += nwaves) {
    const int b = ro[u], e = ro[u + 1], tb = t_ro[u], te = t_ro[u + 1];
    unsigned long long acc = 0ull;
    if (e - b == te - tb) {
      for (int k = b + lane; k < e; k += 64) acc += sym_mix((unsigned long long)(unsigned)ci[k]);
      for (int k = tb + lane; k < te; k += 64) acc -= sym_mix((unsigned long long)(unsigned)t_ci[k]);
    } else {
      acc = 1ull;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0 && acc != 0ull) *bad = 1;
  }
}

}  // namespace grx

using namespace grx;

grx_status_t grx::graph_is_symmetric(grx_context_t ctx, grx_graph_t g, bool* result) {
  std::lock_guard<std::recursive_mutex> lk(g->prep_mu);
  if (g->sym_checked == 0) {
    grx_status_t st = graph_build_transpose(ctx, g);  // needed anyway when the answer is "no"
    if (st != GRX_SUCCESS) return st;
    const int32_t V = g->V;
    hipStream_t s = ctx->stream;
    prep_timer tm("symmetry check (row hashes, CSR against transpose)", s);
    dev_scratch bad_buf;
    GRX_HIP(bad_buf.alloc(sizeof(int32_t)));
    int32_t* bad = bad_buf.as<int32_t>();
    GRX_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(sym_compare_rows_kernel, dim3(2048), dim3(256), 0, s, g->ro, g->ci, g->t_ro, g->t_ci, V, bad);
    int32_t hb = 0;
    GRX_HIP(hipMemcpyAsync(&hb, bad, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    GRX_HIP(hipStreamSynchronize(s));
    GRX_HIP(hipGetLastError());
    g->sym_checked = hb ? 2 : 1;
  }
  *result = g->sym_checked == 1;
  return GRX_SUCCESS;
}

grx_status_t grx::graph_build_transpose(grx_context_t ctx, grx_graph_t g) {
  // (round 5) under the handle's build lock, into locals, published together at the end: a second context that makes its
  // first search meanwhile finds either nothing or the finished transpose, and an error return leaves nothing behind.
  std::lock_guard<std::recursive_mutex> lk(g->prep_mu);
  if (g->has_transpose) return GRX_SUCCESS;
  const int32_t V = g->V;
  const int64_t E = g->E;
  hipStream_t s = ctx->stream;
  prep_timer tm("transpose (expand + radix sort + offsets)", s);
  const bool weighted = g->w != nullptr;
  if (V >= (1 << 29)) return fail(GRX_ERROR_UNSUPPORTED, "transpose: more than 2^29 vertices");
  // test aid: GRX_TR_FAIL_ALLOC=1 makes the build fail as an out-of-memory condition would (what callers fall back on)
  if (const char* e = getenv("GRX_TR_FAIL_ALLOC"); e && *e == '1') return fail(GRX_ERROR_OUT_OF_MEMORY, "transpose: scratch for the sort (forced)");
  dev_scratch t_ro, t_ci, t_w;
  if (t_ro.alloc(((size_t)V + 2) * sizeof(int32_t)) != hipSuccess) {
    (void)hipGetLastError();
    return fail(GRX_ERROR_OUT_OF_MEMORY, "transpose: offsets");
  }
  if (E <= 0) {
    GRX_HIP(hipMemsetAsync(t_ro.p, 0, ((size_t)V + 2) * sizeof(int32_t), s));
    GRX_HIP(t_ci.alloc(sizeof(int32_t)));
    if (weighted) GRX_HIP(t_w.alloc(sizeof(float)));
    GRX_HIP(hipStreamSynchronize(s));
  } else {
    // STABLE radix sort of the edges by (destination, hub tier of the source): no atomic decides an order, so the
    // transpose -- and every fp32 sum taken over a column -- is the same on every run and every handle (grx_sort.hpp)
    sort_buffers sb;  // (frees what is not adopted below when it goes out of scope)
    {
      prep_timer t0("  transpose: scratch allocation", s);
      if (sb.alloc(E, weighted) != hipSuccess) {
        (void)hipGetLastError();
        return fail(GRX_ERROR_OUT_OF_MEMORY, "transpose: scratch for the sort");
      }
    }
    const int32_t mean = (int32_t)std::max<int64_t>(1, E / std::max(1, V));
    const bool tiers = getenv("GRX_TR_NOTIERS") == nullptr;
    {
      prep_timer t1("  transpose: expand", s);
      const tr_emit em{g->ro, g->w, tiers ? 16 * mean : INT32_MAX, tiers ? mean : INT32_MAX, sb.keys[0], sb.vals[0], sb.vals2[0]};
      hipLaunchKernelGGL((edge_expand_kernel<tr_emit>), dim3((unsigned)((E + SORT_TILE - 1) / SORT_TILE)), dim3(SORT_BLOCK), 0, s,
                         g->ro, g->ci, V, E, em);
    }
    int res;
    {
      prep_timer t2("  transpose: radix sort", s);
      res = radix_sort_pairs(s, sb, bits_for((uint64_t)V) + 2);
    }
    hipLaunchKernelGGL(sort_boundaries_kernel, dim3(2048), dim3(256), 0, s, sb.keys[res], E, 2, V, t_ro.as<int32_t>());
    GRX_HIP(hipStreamSynchronize(s));
    GRX_HIP(hipGetLastError());
    t_ci.p = sb.vals[res];  // the sorted sources ARE the column array
    if (weighted) t_w.p = sb.vals2[res];
    sb.release(t_ci.p, t_w.p);
  }
  // (the host copy of the offsets is taken by the one consumer that needs it: PageRank's static partition)
  g->t_ro = reinterpret_cast<int32_t*>(t_ro.release());
  g->t_ci = reinterpret_cast<int32_t*>(t_ci.release());
  g->t_w = reinterpret_cast<float*>(t_w.release());
  g->has_transpose = true;
  return GRX_SUCCESS;
}

// ---- hub-first copy of the IN-ROWS a partition brings (round 6) --------------------------------------------------
// A bottom-up level stops at the first in-neighbour it finds in the frontier, so the transpose above lists a vertex's hub
// sources first.  A rank of a partitioned graph holds the in-rows of its slice as the caller built them (generator / file
// order) and does not know the out-degree of a source it does not own -- but a source with d out-edges appears in about
// d / P of a slice's in-edges, so its FREQUENCY in the local column array ranks it just as well.  Same tiers (>= 16 x the
// mean, >= the mean, rest), same stable sort by (row, tier): the rows keep their offsets, only the order inside a row
// changes, identically on every run.  Measured before it existed (profiles/r6_c3_part_sim_twitter.txt): a bottom-up level of
// the 21 M-vertex stand-in took ONE RANK OF EIGHT 0.30-0.35 ms against 0.08 ms for the whole graph on one GPU.
__global__ void hf_count_kernel(const int32_t* __restrict__ ci, int64_t E, int32_t V, int32_t* cnt) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const unsigned u = (unsigned)ci[e];
    if (u < (unsigned)V) atomicAdd(&cnt[u], 1);
  }
}
struct hf_emit {
  const int32_t* cnt;
  int32_t V, hub, mean;
  uint32_t* keys;
  uint32_t* vals;
  __device__ __forceinline__ void operator()(int64_t e, int row, int col) const {
    const int c = (unsigned)col < (unsigned)V ? cnt[col] : 0;
    const uint32_t tier = c >= hub ? 0u : (c >= mean ? 1u : 2u);
    keys[e] = ((uint32_t)row << 2) | tier;
    vals[e] = (uint32_t)col;
  }
};

grx_status_t grx::graph_build_hub_first(grx_context_t ctx, grx_graph_t g) {
  std::lock_guard<std::recursive_mutex> lk(g->prep_mu);
  if (g->hf_state != 0) return GRX_SUCCESS;
  lazy_state state(&g->hf_state);
  const int32_t V = g->V;
  const int64_t E = g->E;
  hipStream_t s = ctx->stream;
  prep_timer tm("partition: hub-first in-rows (count + radix sort)", s);
  if (V <= 0 || E <= 0 || V >= (1 << 29) || getenv("GRX_PART_HUB_FIRST_OFF") != nullptr) return state.done(2);
  sort_buffers sb;
  dev_scratch cnt;
  if (sb.alloc(E, false) != hipSuccess || cnt.alloc((size_t)V * sizeof(int32_t)) != hipSuccess) {
    (void)hipGetLastError();
    return state.done(2);  // (no room for the copy: the rows are probed in the order they came in)
  }
  GRX_HIP(hipMemsetAsync(cnt.p, 0, (size_t)V * sizeof(int32_t), s));
  hipLaunchKernelGGL(hf_count_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, s, g->ci, E, V, cnt.as<int32_t>());
  const int32_t mean = (int32_t)std::max<int64_t>(1, E / std::max(1, V));
  const hf_emit em{cnt.as<int32_t>(), V, 16 * mean, mean, sb.keys[0], sb.vals[0]};
  hipLaunchKernelGGL((edge_expand_kernel<hf_emit>), dim3((unsigned)((E + SORT_TILE - 1) / SORT_TILE)), dim3(SORT_BLOCK), 0, s,
                     g->ro, g->ci, V, E, em);
  const int res = radix_sort_pairs(s, sb, bits_for((uint64_t)V) + 2);
  GRX_HIP(hipStreamSynchronize(s));
  GRX_HIP(hipGetLastError());
  g->hf_ci = reinterpret_cast<int32_t*>(sb.vals[res]);
  sb.release(g->hf_ci);
  return state.done(1);
}

// Test hook (tests/test_sort_gpu.py): the stable radix sort of grx_sort.hpp on caller arrays, in place.
extern "C" grx_status_t grx_debug_radix_sort(grx_context_t ctx, uint32_t* d_keys, uint32_t* d_vals, uint32_t* d_vals2, int64_t n,
                                             int32_t key_bits) {
  if (!ctx || (n > 0 && (!d_keys || !d_vals)) || key_bits < 1 || key_bits > 32 || n < 0)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_debug_radix_sort: bad argument");
  if (n == 0) return GRX_SUCCESS;
  GRX_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  sort_buffers sb;
  if (sb.alloc(n, d_vals2 != nullptr) != hipSuccess) {
    sb.release();
    (void)hipGetLastError();
    return fail(GRX_ERROR_OUT_OF_MEMORY, "grx_debug_radix_sort: scratch");
  }
  const size_t bytes = (size_t)n * sizeof(uint32_t);
  GRX_HIP(hipMemcpyAsync(sb.keys[0], d_keys, bytes, hipMemcpyDeviceToDevice, s));
  GRX_HIP(hipMemcpyAsync(sb.vals[0], d_vals, bytes, hipMemcpyDeviceToDevice, s));
  if (d_vals2) GRX_HIP(hipMemcpyAsync(sb.vals2[0], d_vals2, bytes, hipMemcpyDeviceToDevice, s));
  const int res = radix_sort_pairs(s, sb, key_bits);
  GRX_HIP(hipMemcpyAsync(d_keys, sb.keys[res], bytes, hipMemcpyDeviceToDevice, s));
  GRX_HIP(hipMemcpyAsync(d_vals, sb.vals[res], bytes, hipMemcpyDeviceToDevice, s));
  if (d_vals2) GRX_HIP(hipMemcpyAsync(d_vals2, sb.vals2[res], bytes, hipMemcpyDeviceToDevice, s));
  GRX_HIP(hipStreamSynchronize(s));
  sb.release();
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

// format::csr_t::from_coo on the DEVICE.  Upstream builds a CSR from COO triples with a host counting sort by row
// (formats/csr.hxx:81-140): stable, so the entries of a row keep their input order -- duplicates and self loops included; that
// order is part of the contract (it decides the order of a frontier).  Here: the stable radix sort of grx_sort.hpp by the row
// index, columns and values riding along, then the offsets from the sorted keys.  The result is byte-identical to the host
// builder's (grx_host_csr_from_coo) on the same triples.
extern "C" grx_status_t grx_csr_from_coo_device(grx_context_t ctx, int32_t n_rows, int64_t nnz, const int32_t* d_row_indices,
                                                const int32_t* d_column_indices, const float* d_values, int32_t* d_row_offsets,
                                                int32_t* d_out_columns, float* d_out_values) {
  if (!ctx || n_rows < 0 || nnz < 0 || nnz > 0x7fffffffll || !d_row_offsets ||
      (nnz > 0 && (!d_row_indices || !d_column_indices || !d_out_columns)) || ((d_values == nullptr) != (d_out_values == nullptr)))
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_csr_from_coo_device: bad argument");
  GRX_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (nnz == 0) {
    GRX_HIP(hipMemsetAsync(d_row_offsets, 0, ((size_t)n_rows + 1) * sizeof(int32_t), s));
    GRX_HIP(hipStreamSynchronize(s));
    return GRX_SUCCESS;
  }
  if (n_rows == 0) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_csr_from_coo_device: entries without rows");
  sort_buffers sb;
  if (sb.alloc(nnz, d_values != nullptr) != hipSuccess) {
    sb.release();
    (void)hipGetLastError();
    return fail(GRX_ERROR_OUT_OF_MEMORY, "grx_csr_from_coo_device: scratch");
  }
  const size_t bytes = (size_t)nnz * sizeof(uint32_t);
  GRX_HIP(hipMemcpyAsync(sb.keys[0], d_row_indices, bytes, hipMemcpyDeviceToDevice, s));
  GRX_HIP(hipMemcpyAsync(sb.vals[0], d_column_indices, bytes, hipMemcpyDeviceToDevice, s));
  if (d_values) GRX_HIP(hipMemcpyAsync(sb.vals2[0], d_values, bytes, hipMemcpyDeviceToDevice, s));
  // all 32 key bits when a row index may lie outside [0, n_rows): such a key then sorts to the end, where it is seen
  const int res = radix_sort_pairs(s, sb, 32);
  uint32_t last = 0u;
  GRX_HIP(hipMemcpyAsync(&last, sb.keys[res] + (nnz - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  hipLaunchKernelGGL(sort_boundaries_kernel, dim3(ctx->num_cus * 4), dim3(256), 0, s, sb.keys[res], nnz, 0, n_rows, d_row_offsets);
  GRX_HIP(hipMemcpyAsync(d_out_columns, sb.vals[res], bytes, hipMemcpyDeviceToDevice, s));
  if (d_values) GRX_HIP(hipMemcpyAsync(d_out_values, sb.vals2[res], bytes, hipMemcpyDeviceToDevice, s));
  GRX_HIP(hipStreamSynchronize(s));
  sb.release();
  GRX_HIP(hipGetLastError());
  if (last >= (uint32_t)n_rows) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_csr_from_coo_device: a row index lies outside [0, n_rows)");
  return GRX_SUCCESS;
}
