// grx_api.hip -- context, graph view, scratch arena and statistics of the C ABI.
#include "grx_engine.hpp"

#include <algorithm>
#include "grx_bin.hpp"
#include "grx_mid.hpp"

#include <mutex>

namespace grx {

static thread_local std::string g_error;

void set_error(const std::string& msg) { g_error = msg; }
grx_status_t fail(grx_status_t code, const std::string& msg) {
  g_error = msg;
  return code;
}

__global__ void fill_i32_kernel(int32_t* p, int32_t value, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // 16-byte stores on the aligned body
  int64_t n4 = n / 4;
  int4 v4 = make_int4(value, value, value, value);
  int4* p4 = reinterpret_cast<int4*>(p);
  for (int64_t j = i; j < n4; j += stride) p4[j] = v4;
  for (int64_t j = n4 * 4 + i; j < n; j += stride) p[j] = value;
}

__global__ void fill_f32_kernel(float* p, float value, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t n4 = n / 4;
  float4 v4 = make_float4(value, value, value, value);
  float4* p4 = reinterpret_cast<float4*>(p);
  for (int64_t j = i; j < n4; j += stride) p4[j] = v4;
  for (int64_t j = n4 * 4 + i; j < n; j += stride) p[j] = value;
}

static int fill_grid(int64_t n) {
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  return (int)blocks;
}

hipError_t fill_i32(hipStream_t s, int32_t* p, int32_t value, int64_t n) {
  if (n <= 0) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(p) & 15) != 0)  // caller buffer not 16-byte aligned
    return hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p), value, (size_t)n, s);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(fill_grid(n)), dim3(256), 0, s, p, value, n);
  return hipGetLastError();
}

hipError_t fill_f32(hipStream_t s, float* p, float value, int64_t n) {
  if (n <= 0) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) {
    int bits;
    memcpy(&bits, &value, 4);
    return hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p), bits, (size_t)n, s);
  }
  hipLaunchKernelGGL(fill_f32_kernel, dim3(fill_grid(n)), dim3(256), 0, s, p, value, n);
  return hipGetLastError();
}

// Sampled content fingerprint of a device CSR (grx_csr_fingerprint).  <<<1, 1024>>>
__global__ void fingerprint_kernel(const int32_t* ro, const int32_t* ci, const float* w, int32_t V, int32_t E,
                                   unsigned long long* out) {
  auto mix = [](unsigned long long x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
  };
  const int t = threadIdx.x;
  unsigned long long acc = 0ull;
  // 1024 evenly spaced samples of each array (every element when it is shorter), position-keyed
  const long long nr = (long long)V + 1, ne = (long long)E;
  const long long ir = nr > 1024 ? (long long)t * nr / 1024 : t;
  if (ir < nr) acc += mix(((unsigned long long)ir << 32) ^ (unsigned)ro[ir]);
  const long long ie = ne > 1024 ? (long long)t * ne / 1024 : t;
  if (ie < ne) {
    acc += mix((1ull << 63) ^ ((unsigned long long)ie << 32) ^ (unsigned)ci[ie]);
    if (w) acc += mix((1ull << 62) ^ ((unsigned long long)ie << 32) ^ (unsigned)__float_as_int(w[ie]));
  }
  if (t == 0) acc += mix((unsigned long long)(unsigned)ro[V] ^ 0xabcdull);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((t & 63) == 0) atomicAdd(out, acc);
}

// FULL content hash of one 32-bit array (grx_csr_hash): a position-keyed 64-bit mix of every element, summed (the sum is
// order independent, so any grid works).  One streaming pass, 16-byte loads when the base is aligned.
__device__ __forceinline__ unsigned long long hash_mix(unsigned long long pos, unsigned v) {
  unsigned long long x = (pos * 0x9e3779b97f4a7c15ull) ^ ((unsigned long long)v * 0xc2b2ae3d27d4eb4full);
  x = (x ^ (x >> 29)) * 0xbf58476d1ce4e5b9ull;
  return x ^ (x >> 32);
}
__global__ __launch_bounds__(256) void content_hash_kernel(const unsigned* __restrict__ p, long long n, unsigned long long salt,
                                                           unsigned long long* out) {
  unsigned long long acc = 0ull;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
  if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
    const uint4* p4 = reinterpret_cast<const uint4*>(p);
    const long long n4 = n >> 2;
    for (long long i = tid; i < n4; i += nt) {
      const uint4 v = p4[i];
      const unsigned long long b = salt + 4ull * (unsigned long long)i;
      acc += hash_mix(b, v.x) + hash_mix(b + 1, v.y) + hash_mix(b + 2, v.z) + hash_mix(b + 3, v.w);
    }
    for (long long i = (n4 << 2) + tid; i < n; i += nt) acc += hash_mix(salt + (unsigned long long)i, p[i]);
  } else {
    for (long long i = tid; i < n; i += nt) acc += hash_mix(salt + (unsigned long long)i, p[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ unsigned long long s_part[4];
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

grx_status_t pipeline_prepare(grx_context_t ctx, grx_graph_t g, pipe_args* a) {
  const size_t V = (size_t)g->V, E = (size_t)g->E;
  const size_t grid = (size_t)advance_grid(ctx);
  // The multi-level body (grx_mid.hpp, second version -- the only one a default build carries) lays MID_WGS private
  // regions of MID_SEG entries plus an overflow area of up to V entries over each parity buffer, whatever the grid:
  // size for it unconditionally (2 MB per buffer), so that a device or partition with few CUs cannot overrun
  // (the round-2 guard fell back to a body that is not compiled in).
  const size_t mid_tiles = std::max((size_t)MID_WGS * MID_SEG_TILES + V / TILE + 2, ((size_t)MID_OVF_BASE + V + 2 * TILE) / TILE);
  const size_t max_tiles = std::max(V / TILE + grid * (TILE_RESERVE + 1) + 8, mid_tiles);
  const size_t max_chunks = E / CHUNK + max_tiles + 8;
  for (int i = 0; i < 2; ++i) GRX_HIP(ctx->frontier[i].reserve(max_tiles * TILE * sizeof(int32_t)));
  GRX_HIP(ctx->tile_chunks.reserve(max_tiles * sizeof(int32_t)));
  GRX_HIP(ctx->tile_sums.reserve(max_tiles * sizeof(int32_t)));
  GRX_HIP(ctx->tile_count.reserve(max_tiles * sizeof(int32_t)));
  GRX_HIP(ctx->chunk_tile.reserve(max_chunks * 2 * sizeof(int32_t)));
  a->ro = g->ro;
  a->ci = g->ci;
  a->w = g->w;
  a->V = g->V;
  a->ctrl = ctx->d_ctrl;
  a->mailbox = ctx->d_mailbox;
  a->frontier[0] = ctx->frontier[0].as<int32_t>();
  a->frontier[1] = ctx->frontier[1].as<int32_t>();
  a->tile_chunks = ctx->tile_chunks.as<int32_t>();
  a->tile_sums = ctx->tile_sums.as<int32_t>();
  a->tile_count = ctx->tile_count.as<int32_t>();
  a->chunk_tile = ctx->chunk_tile.as<int32_t>();
  a->bu_part = nullptr;
  // grx_mid.hpp: [first version: 2 x MID_AUX_CAP int2][second version: 2 x MID_AUX2_CAP int4][flag words]
  const size_t aux1 = (size_t)2 * MID_AUX_CAP * 2 * sizeof(int32_t);
  const size_t aux2 = (size_t)2 * MID_AUX2_CAP * 4 * sizeof(int32_t);
  GRX_HIP(ctx->mid_aux.reserve(aux1 + aux2 + MID_FLAG_WORDS * sizeof(unsigned long long)));
  a->mid_aux = ctx->mid_aux.ptr;
  a->mid_aux2 = static_cast<char*>(ctx->mid_aux.ptr) + aux1;
  a->mid_flags = reinterpret_cast<unsigned long long*>(static_cast<char*>(ctx->mid_aux.ptr) + aux1 + aux2);
  // the second version lays MID_WGS private regions + an overflow area of up to V entries over a parity buffer
  const char* mv = getenv("GRX_MID_VERSION");
  a->mid_version = (mv && *mv == '1') ? 1 : 2;  // honoured by builds that carry both bodies (grx_mid.hpp)
  a->mid_seg_cap = MID_SEG;
  a->mid_exit_v = MID_EXIT_V;
  a->mid_exit_e = MID_EXIT_E;
  a->mid_hub_deg = 16;
  a->mid_refill_max = 0;
  if (const char* e = getenv("GRX_MID_HUB_DEG")) a->mid_hub_deg = atoi(e);
  if (const char* e = getenv("GRX_MID_EXIT_E")) { const int x = atoi(e); if (x >= 1) a->mid_exit_e = x; }
  if (const char* e = getenv("GRX_MID_SEG_CAP")) { const int x = atoi(e); if (x >= 0 && x <= MID_SEG) a->mid_seg_cap = x; }    // test knobs
  if (const char* e = getenv("GRX_MID_EXIT_V")) { const int x = atoi(e); if (x >= 1 && x <= MID_EXIT_V) a->mid_exit_v = x; }
  const char* md = getenv("GRX_MID_DEBUG");
  if (md && *md == '1') a->mid_version |= 0x100;  // per-phase clock sums in ctrl.spare (grx_debug_ctrl)
  static_assert(MID_OVF_BASE == MID_WGS * MID_SEG, "regions first, overflow area behind them");
  return GRX_SUCCESS;
}

#ifndef GRX_WITH_BLOCK
// The default library does not carry grx_block.hip (gunrock_amd/build.py): the searches take their level-synchronous paths.
grx_status_t blk_prepare(grx_context_t, grx_graph_t, bool, bool* usable) {
  if (usable) *usable = false;
  return GRX_SUCCESS;
}
grx_status_t blk_search(grx_context_t, grx_graph_t, int32_t, const grx_options_t&, bool, void*, float*) {
  return fail(GRX_ERROR_UNSUPPORTED, "block-asynchronous relaxation is not part of this build (python -m gunrock_amd.build --with-block)");
}
void blk_graph_free(void*) {}
#endif

}  // namespace grx

using namespace grx;

#ifndef GRX_WITH_BLOCK
extern "C" grx_status_t grx_debug_block_search_host(grx_host_csr_t, int32_t, int32_t, int32_t, uint32_t, uint32_t*, grx_block_stats_t*) {
  return fail(GRX_ERROR_UNSUPPORTED, "grx_debug_block_search_host: not part of this build (python -m gunrock_amd.build --with-block)");
}
#endif

extern "C" {

const char* grx_last_error_string(void) { return g_error.c_str(); }
const char* grx_version_string(void) { return "grx 0.1 (gfx950)"; }

void grx_options_default(grx_options_t* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->advance_load_balance = GRX_LB_BLOCK_MAPPED;
  o->filter_algorithm = GRX_FILTER_PREDICATED;
  o->enable_filter = 0;
  o->enable_uniquify = 0;
  o->uniquify_algorithm = GRX_UNIQUIFY_UNIQUE;
  o->best_effort_uniquify = 1;
  o->uniquify_percent = 100.0f;
}

grx_status_t grx_context_create(int32_t device, void* stream, grx_context_t* out) {
  if (!out) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_context_create: null out");
  int count = 0;
  GRX_HIP(hipGetDeviceCount(&count));
  if (device < 0 || device >= count)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_context_create: no such device");
  GRX_HIP(hipSetDevice(device));
  grx_context* c = new grx_context();
  c->device = device;
  if (stream) {
    c->stream = reinterpret_cast<hipStream_t>(stream);
    c->own_stream = false;
  } else {
    GRX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  GRX_HIP(hipEventCreate(&c->ev_begin));
  GRX_HIP(hipEventCreate(&c->ev_end));
  hipDeviceProp_t prop;
  GRX_HIP(hipGetDeviceProperties(&prop, device));
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wall_clock_khz = (double)khz;
  }
  GRX_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_ctrl), sizeof(ctrl_t)));
  // NOT hipMemset: that is queued on the null stream and may be submitted much later than
  // work on this context's (non-blocking) stream -- it once zeroed a running search
  GRX_HIP(hipMemsetAsync(c->d_ctrl, 0, sizeof(ctrl_t), c->stream));
  GRX_HIP(hipStreamSynchronize(c->stream));
  {
    // which XCDs does this device (or partition) have?  Every workgroup of a wide launch reports the
    // hardware XCC id of its CU; kernels that give an XCD exclusive ownership of data (grx_bin.hpp)
    // map the ids to dense indices through this mask.
    unsigned* d_mask = reinterpret_cast<unsigned*>(c->d_ctrl);
    hipLaunchKernelGGL(xcc_census_kernel, dim3(c->num_cus * 16), dim3(64), 0, c->stream, d_mask);
    unsigned h_mask = 0;
    GRX_HIP(hipMemcpyAsync(&h_mask, d_mask, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    GRX_HIP(hipStreamSynchronize(c->stream));
    GRX_HIP(hipMemsetAsync(c->d_ctrl, 0, sizeof(ctrl_t), c->stream));
    GRX_HIP(hipStreamSynchronize(c->stream));
    if (h_mask == 0) h_mask = 1u;
    c->xcc_mask = h_mask;
    c->n_xcd = __builtin_popcount(h_mask);
  }
  GRX_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_ctrl), sizeof(ctrl_t), hipHostMallocDefault));
  memset(c->h_ctrl, 0, sizeof(ctrl_t));
  void* mb = nullptr;
  GRX_HIP(hipHostMalloc(&mb, 64, hipHostMallocMapped));
  memset(mb, 0, 64);
  c->h_mailbox = reinterpret_cast<volatile int32_t*>(mb);
  void* dmb = nullptr;
  GRX_HIP(hipHostGetDevicePointer(&dmb, mb, 0));
  c->d_mailbox = reinterpret_cast<int32_t*>(dmb);
  *out = c;
  return GRX_SUCCESS;
}

grx_status_t grx_context_synchronize(grx_context_t ctx) {
  if (!ctx) return fail(GRX_ERROR_INVALID_ARGUMENT, "null context");
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  return GRX_SUCCESS;
}

void* grx_context_stream(grx_context_t ctx) { return ctx ? (void*)ctx->stream : nullptr; }

grx_status_t grx_context_order_after(grx_context_t ctx, void* producer_stream) {
  if (!ctx) return fail(GRX_ERROR_INVALID_ARGUMENT, "null context");
  hipStream_t ps = reinterpret_cast<hipStream_t>(producer_stream);
  if (ps == ctx->stream) return GRX_SUCCESS;
  GRX_HIP(hipSetDevice(ctx->device));  // the event below belongs to the context's device, whatever the caller's current one is
  const hipError_t q = hipStreamQuery(ps);
  if (q == hipSuccess) return GRX_SUCCESS;  // idle: nothing to be ordered after
  // anything but "idle" -- still busy, or a stream under capture (the query itself is an error there) -- : record and wait
  if (q != hipErrorNotReady) (void)hipGetLastError();
  if (!ctx->ev_order) GRX_HIP(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
  GRX_HIP(hipEventRecord(ctx->ev_order, ps));
  GRX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_order, 0));
  return GRX_SUCCESS;
}

grx_status_t grx_context_destroy(grx_context_t ctx) {
  if (!ctx) return GRX_SUCCESS;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& b : ctx->frontier) b.release();
  ctx->tile_chunks.release();
  ctx->tile_sums.release();
  ctx->tile_count.release();
  ctx->bu_part.release();
  for (auto& b : ctx->far) b.release();
  ctx->chunk_tile.release();
  for (auto& b : ctx->bitmap) b.release();
  ctx->labels.release();
  for (auto& b : ctx->fbuf) b.release();
  ctx->misc.release();
  ctx->mid_aux.release();
  ctx->bins.release();
  ctx->bin_fill.release();
  for (auto& b : ctx->blk_buf) b.release();
  for (auto& b : ctx->rbins) b.release();
  if (ctx->d_ctrl) (void)hipFree(ctx->d_ctrl);
  if (ctx->h_ctrl) (void)hipHostFree(ctx->h_ctrl);
  if (ctx->h_mailbox) (void)hipHostFree((void*)ctx->h_mailbox);
  if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
  if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
  if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return GRX_SUCCESS;
}

grx_status_t grx_graph_create_csr(grx_context_t ctx, int32_t V, int32_t E, const int32_t* ro,
                                  const int32_t* ci, const float* w, int32_t directed,
                                  int32_t weighted, int32_t symmetric, grx_graph_t* out) {
  if (!ctx || !out) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_graph_create_csr: null argument");
  if (V < 0 || E < 0 || !ro || (E > 0 && !ci))
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_graph_create_csr: bad CSR arrays");
  grx_graph* g = new grx_graph();
  g->ctx = ctx;
  g->V = V;
  g->E = E;
  g->ro = ro;
  g->ci = ci;
  g->w = w;
  g->directed = directed;
  g->weighted = weighted;
  g->symmetric = symmetric;
  *out = g;
  return GRX_SUCCESS;
}

grx_status_t grx_graph_destroy(grx_graph_t g) {
  if (!g) return GRX_SUCCESS;
  if (g->t_ro) (void)hipFree(g->t_ro);
  if (g->t_ci) (void)hipFree(g->t_ci);
  if (g->t_w) (void)hipFree(g->t_w);
  if (g->closed0) (void)hipFree(g->closed0);
  if (g->bu_heads) (void)hipFree(g->bu_heads);
  if (g->hf_ci) (void)hipFree(g->hf_ci);
  if (g->bin_off) (void)hipFree(g->bin_off);
  if (g->rb_off) (void)hipFree(g->rb_off);
  if (g->rb_g2b16) (void)hipFree(g->rb_g2b16);
  if (g->bin_tab8) (void)hipFree(g->bin_tab8);
  for (void* b : g->blk) blk_graph_free(b);
  if (g->pr_blocks) (void)hipFree(g->pr_blocks);
  if (g->pr_piece) (void)hipFree(g->pr_piece);
  if (g->pr_long) (void)hipFree(g->pr_long);
  if (g->xb_ro) (void)hipFree(g->xb_ro);
  if (g->xb_pos) (void)hipFree(g->xb_pos);
  if (g->xb_ci) (void)hipFree(g->xb_ci);
  if (g->xb_w) (void)hipFree(g->xb_w);
  if (g->xb_blocks) (void)hipFree(g->xb_blocks);
  if (g->xb_piece) (void)hipFree(g->xb_piece);
  if (g->xb_long) (void)hipFree(g->xb_long);
  if (g->xb_perm) (void)hipFree(g->xb_perm);
  delete g;
  return GRX_SUCCESS;
}

grx_status_t grx_csr_fingerprint(grx_context_t ctx, int32_t V, int32_t E, const int32_t* ro, const int32_t* ci,
                                 const float* w, uint64_t* out) {
  if (!ctx || !ro || !out || V < 0 || E < 0 || (E > 0 && !ci))
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_csr_fingerprint: bad argument");
  GRX_HIP(hipSetDevice(ctx->device));
  GRX_HIP(ctx->misc.reserve(64));
  unsigned long long* d = ctx->misc.as<unsigned long long>();
  GRX_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(fingerprint_kernel, dim3(1), dim3(1024), 0, ctx->stream, ro, ci, w, V, E, d);
  unsigned long long h = 0;
  GRX_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  *out = (uint64_t)h;
  return GRX_SUCCESS;
}

grx_status_t grx_csr_hash(grx_context_t ctx, int32_t V, int32_t E, const int32_t* ro, const int32_t* ci, const float* w,
                          uint64_t* out) {
  if (!ctx || !ro || !out || V < 0 || E < 0 || (E > 0 && !ci))
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_csr_hash: bad argument");
  GRX_HIP(hipSetDevice(ctx->device));
  GRX_HIP(ctx->misc.reserve(64));
  unsigned long long* d = ctx->misc.as<unsigned long long>();
  hipStream_t s = ctx->stream;
  GRX_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), s));
  auto pass = [&](const void* p, long long n, unsigned long long salt) {
    if (!p || n <= 0) return;
    const long long blocks = std::min<long long>((n / 4 + 255) / 256 + 1, (long long)ctx->num_cus * 8);
    hipLaunchKernelGGL(content_hash_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const unsigned*>(p), n, salt, d);
  };
  pass(ro, (long long)V + 1, 0x1000000000000000ull);
  pass(ci, (long long)E, 0x2000000000000000ull);
  pass(w, (long long)E, 0x3000000000000000ull);
  unsigned long long h = 0;
  GRX_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, s));
  GRX_HIP(hipStreamSynchronize(s));
  GRX_HIP(hipGetLastError());
  // (V, E, presence of weights) are part of the identity a caller keys on; fold them in so that a hash is never 0 by accident
  *out = (uint64_t)(h ^ ((unsigned long long)(unsigned)V << 32) ^ (unsigned long long)(unsigned)E ^ (w ? 0x8000000000000000ull : 0ull));
  return GRX_SUCCESS;
}

int32_t grx_graph_number_of_vertices(grx_graph_t g) { return g ? g->V : 0; }
int32_t grx_graph_number_of_edges(grx_graph_t g) { return g ? g->E : 0; }

grx_status_t grx_get_run_stats(grx_context_t ctx, grx_run_stats_t* out) {
  if (!ctx || !out) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_get_run_stats: null argument");
  *out = ctx->stats;
  return GRX_SUCCESS;
}

/* 1: this build of the library carries the block-asynchronous relaxation for road-like graphs (grx_block.hip, -DGRX_WITH_BLOCK:
 * `python -m gunrock_amd.build --with-block`), 0: it does not (the default since round 6) and GRX_BLOCK=1 has no effect */
int32_t grx_has_block_async(void) {
#ifdef GRX_WITH_BLOCK
  return 1;
#else
  return 0;
#endif
}

grx_status_t grx_get_block_stats(grx_context_t ctx, grx_block_stats_t* out) {
  if (!ctx || !out) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_get_block_stats: null argument");
  *out = ctx->block_stats;
  return GRX_SUCCESS;
}

grx_status_t grx_get_level_profile(grx_context_t ctx, grx_level_profile_t* out, int32_t capacity,
                                   int32_t* n_levels) {
  if (!ctx || !n_levels) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_get_level_profile: null argument");
  const int32_t n = (int32_t)ctx->levels.size();
  *n_levels = n;
  for (int32_t i = 0; i < n && i < capacity && out; ++i) {
    out[i].frontier_size = ctx->levels[i].frontier_size;
    out[i].edges = ctx->levels[i].edges;
    out[i].advance_ms = ctx->levels[i].advance_ms;
    out[i].other_ms = ctx->levels[i].other_ms;
    out[i].bottom_up = ctx->levels[i].bottom_up;
    out[i].reserved = 0;
    out[i].bu_open = ctx->levels[i].bu_open;
    out[i].bu_probes = ctx->levels[i].bu_probes;
  }
  return GRX_SUCCESS;
}

}  // extern "C"
