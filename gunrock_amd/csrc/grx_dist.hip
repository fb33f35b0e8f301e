// grx_dist.hip -- partitioned (multi-GPU) BFS enactor: device side + level-group C ABI.
//
// The reference is single-GPU only: every operator throws when
// `context.size() != 1` (advance/advance.hxx:129-132, filter/filter.hxx:96-99) and no
// NCCL/RCCL/MPI call exists in its tree.  This is the MI355X design:
//
//   * one process per GPU; rank r owns the vertex slice [r * S, min((r + 1) * S, V)), S a
//     multiple of 2048, the OUT-rows of that slice (global column ids) and -- for the
//     bottom-up step -- its IN-rows (the same rows when the graph is symmetric);
//   * every level moves exactly one fixed-size message per pair of GPUs: an S-bit bitmap.
//       top-down level : bit v of the slice sent to owner(v) = "I discovered v" (deduplicated
//                        against everything this rank ever sent); the owner ORs the P - 1
//                        slices it receives, claims the new ones and appends them to its queue;
//       bottom-up level: every rank sends its frontier slice to everybody (the all-to-all
//                        degenerates to an all-gather), so each rank holds the whole-graph
//                        frontier bitmap and scans the in-edges of its unvisited vertices.
//     Fixed sizes mean NO size exchange and no host round trip: like the single-GPU enactor
//     the host enqueues level groups blindly (kernels + RCCL all_to_all_single + a 4-word
//     all_reduce of the frontier statistics) and reads `done` once per batch.  A slice is
//     S / 8 bytes (0.6 MB for a 4.85 M-vertex slice): ~4 us on one 153 GB/s xGMI link, and on
//     the full mesh every pair has its own link.
//   * the direction (Beamer) is chosen ON THE DEVICE from the all-reduced statistics, so all
//     ranks take the same branch without talking to the host;
//   * `parts = 2` cuts a top-down level in two halves with their own bitmaps: the exchange
//     of the first half runs on the communication stream while the second half is still
//     advancing on the compute stream (gunrock_amd/distributed.py).
#include "grx_engine.hpp"
#include "grx_bfs_kernels.hpp"

#include <rccl/rccl.h>  // types only: the library is opened at run time (rccl_api below)

#include <climits>
#include <dlfcn.h>

namespace grx {

constexpr int DIST_MAX_RANKS = 64;

struct dist_args {
  int32_t n_ranks, my_rank;
  int32_t lo, hi;            // owned vertices
  int32_t slice_words;       // S / 32
  int32_t parts;             // 1 or 2 top-down halves
  unsigned* send;            // [parts][n_ranks][slice_words]
  const unsigned* recv;      // [parts][n_ranks][slice_words]
  unsigned* sent;            // [n_ranks * slice_words]: remote vertices this rank already reported
  long long* stats_local;    // {frontier vertices, frontier out-edges, 0, 0} of the NEXT level (this rank)
  const long long* stats_global;  // the same, summed over ranks (all_reduce by the host library)
  long long e_global;        // edges of the whole graph
  int32_t do_enabled;
};

// Top-down claim on a partitioned graph: owned targets are claimed on the label like the
// single-GPU policy; targets of other ranks are reported once through the outgoing bitmap.
struct bfs_policy_dist {
  using src_state = int;
  int32_t* dist;
  unsigned* sent;
  unsigned* send;
  int lo, hi;
  int next_depth;
  __device__ __forceinline__ void begin(ctrl_t* c) { next_depth = c->level + 1; }
  __device__ __forceinline__ src_state load_source(int) const { return 0; }
  static constexpr bool two_claims = true;
  __device__ __forceinline__ bool owned(int n) const { return n >= lo && n < hi; }
  // ONE load from a selected address (label of an owned target, `sent` word of a remote one)
  // cand: 0 for an owned target, its bit in the 32-vertex word for a remote one
  __device__ __forceinline__ bool precheck(src_state, int n, int, int& cand) const {
    const bool own = owned(n);
    cand = own ? 0 : (int)(1u << (n & 31));
    const unsigned* p = own ? reinterpret_cast<const unsigned*>(dist + n) : sent + (n >> 5);
    const unsigned v = *p;
    return own ? (int)v > next_depth : (v & (1u << (n & 31))) == 0u;
  }
  __device__ __forceinline__ int claim(int n, int) const {
    if (owned(n)) return atomicMin(&dist[n], next_depth);
    return (int)atomicOr(&sent[n >> 5], 1u << (n & 31));
  }
  // a remote target this rank reports for the first time goes into the outgoing bitmap
  __device__ __forceinline__ bool need2(int raw1, int cand) const { return cand != 0 && ((unsigned)raw1 & (unsigned)cand) == 0u; }
  __device__ __forceinline__ int claim2(int n) const { return (int)atomicOr(&send[n >> 5], 1u << (n & 31)); }
  __device__ __forceinline__ int code(int raw1, int, int n, int) const {
    return (owned(n) && next_depth < raw1) ? 1 : 0;
  }
  __device__ __forceinline__ int visit(src_state, int n, int) const {
    const int r1 = claim(n, 0);
    if (owned(n)) return next_depth < r1 ? 1 : 0;
    if (((unsigned)r1 & (1u << (n & 31))) == 0u) (void)claim2(n);
    return 0;
  }
};

__global__ void dist_init_kernel(pipe_args a, dist_args x, int32_t* dist, int src_if_owned) {
  const int tid = threadIdx.x;
  const int src = src_if_owned;
  a.frontier[0][tid] = (tid == 0 && src >= 0) ? src : -1;
  if (tid == 0) {
    ctrl_t* c = a.ctrl;
    const int deg = src >= 0 ? a.ro[src + 1] - a.ro[src] : 0;
    a.tile_sums[0] = deg;
    a.tile_chunks[0] = (deg + CHUNK - 1) / CHUNK;
    a.tile_count[0] = src >= 0 ? 1 : 0;
    c->level = -1;
    c->done = 0;
    c->mode = 0;
    c->n_tiles[0] = src >= 0 ? 1 : 0;
    c->n_tiles[1] = 0;
    c->n_items[0] = src >= 0 ? 1 : 0;
    c->n_items[1] = 0;
    c->q_edges[0] = deg;
    c->q_edges[1] = 0;
    c->total_chunks = 0;
    c->edges_visited = 0;
    c->vertices_visited = 0;
    c->g_edges_visited = 0;
    c->frontier_bitmap = 0;
    c->convert = 0;
    c->bu_R = 0;
    c->bu_T = 0;
    c->bu_open = 0;
    c->bu_probes = 0;
    if (src >= 0) dist[src] = 0;
    x.stats_local[0] = src >= 0 ? 1 : 0;
    x.stats_local[1] = deg;
    x.stats_local[2] = 0;
    x.stats_local[3] = 0;
    a.mailbox[0] = 0;
    a.mailbox[1] = 0;
  }
}

// Head of a level group: global termination + direction choice from the all-reduced
// statistics (identical on every rank), local bookkeeping, chunk map of a top-down level.
// <<<1, 1024>>>
__global__ __launch_bounds__(PLAN_BLOCK) void dist_head_kernel(pipe_args a, dist_args x) {
  __shared__ int s_wave[PLAN_BLOCK / 64 + 1];
  __shared__ unsigned long long s_esum[2];
  ctrl_t* c = a.ctrl;
  const int tid = threadIdx.x;
  if (c->done) return;
  if (tid == 0) {
    s_esum[0] = s_esum[1] = 0ull;
    const int level = c->level + 1;
    const int p = level & 1;
    const long long n_f = x.stats_global[0], m_f = x.stats_global[1];
    if (n_f == 0) {
      c->done = 1;
      c->level = level;
      a.mailbox[1] = level;
      a.mailbox[0] = 1;
    } else {
      const int is_bitmap = c->frontier_bitmap;
      int mode = c->mode;
      if (x.do_enabled) {
        const long long m_u = x.e_global - c->g_edges_visited;
        if (mode == 0) {
          if (m_f > m_u / DO_ALPHA && n_f > 256) mode = 1;
        } else {
          const long long v_global = (long long)x.n_ranks * x.slice_words * 32;
          if (n_f < v_global / DO_BETA) mode = 0;
        }
      }
      c->convert = (mode == 0 && is_bitmap) ? 1 : ((mode == 1 && !is_bitmap) ? 2 : 0);
      c->mode = mode;
      c->level = level;
      c->g_edges_visited += m_f;
      c->edges_visited += c->q_edges[p];      // this rank's share (dist_stats_kernel / init)
      c->vertices_visited += c->n_items[p];
      c->n_tiles[p ^ 1] = 0;
      if (c->convert == 1) {
        c->n_tiles[p] = 0;       // the queue of this level is rebuilt from the bitmap ...
        c->total_chunks = -1;    // ... and walked in tile mode (no chunk map)
      }
      c->frontier_bitmap = mode;
      a.mailbox[1] = level;
    }
  }
  __syncthreads();
  if (c->done || c->mode != 0 || c->convert == 1) return;
  plan_body<PLAN_BLOCK>(a, c, 1, s_wave, s_esum);
}

// Before the exchange.  Bottom-up level: publish this rank's frontier slice to every peer
// (from the labels at the top-down -> bottom-up switch, which also builds the visited
// slice).  Top-down level: clear the outgoing candidate bitmaps; at the switch back,
// rebuild the queue from the frontier slice.
__global__ __launch_bounds__(ADV_BLOCK) void dist_prep_kernel(pipe_args a, dobfs_args d, dist_args x) {
  __shared__ words_smem sm;
  ctrl_t* c = a.ctrl;
  if (c->done) return;
  const int convert = c->convert;
  const int level = c->level;
  const int p = level & 1;
  const int tid = threadIdx.x;
  const int lane = dev::lane_id();
  if (c->mode == 1) {
    unsigned* fcur = d.fbits[p];
    const int n_chunks = d.n_words / 2;
    const int wave = (blockIdx.x * ADV_BLOCK + tid) >> 6;
    const int n_waves = (gridDim.x * ADV_BLOCK) >> 6;
    for (int ch = wave; ch < n_chunks; ch += n_waves) {
      unsigned w0, w1;
      if (convert == 2) {
        const int v = x.lo + ch * 64 + lane;
        const bool in = v < x.hi;
        const int dv = in ? d.dist[v] : INT_MAX;
        const bool no_in = in ? (d.t_ro[v + 1] == d.t_ro[v]) : true;  // can never be found bottom-up
        const unsigned long long vis = dev::ballot(dv != INT_MAX || no_in);
        const unsigned long long fr = dev::ballot(dv == level);
        w0 = (unsigned)fr;
        w1 = (unsigned)(fr >> 32);
        if (lane == 0) {
          d.visited[2 * ch] = (unsigned)vis;
          d.visited[2 * ch + 1] = (unsigned)(vis >> 32);
          fcur[2 * ch] = w0;
          fcur[2 * ch + 1] = w1;
        }
      } else {
        w0 = fcur[2 * ch];
        w1 = fcur[2 * ch + 1];
      }
      if (lane < x.n_ranks) {  // lane j addresses peer j
        unsigned* dst = x.send + (size_t)lane * x.slice_words + 2 * ch;
        dst[0] = w0;
        dst[1] = w1;
      }
    }
    return;
  }
  const size_t total = (size_t)x.parts * x.n_ranks * x.slice_words;
  for (size_t i = (size_t)blockIdx.x * ADV_BLOCK + tid; i < total; i += (size_t)gridDim.x * ADV_BLOCK) x.send[i] = 0u;
  if (convert == 1) {
    const unsigned* fin = d.fbits[p];
    words_to_tiles(a, c, p, d.n_words, x.lo, [fin](int w) { return fin[w]; }, sm);
  }
}

// Top-down half `part` of `parts`: units part, part + parts, ... of the level.
__global__ __launch_bounds__(ADV_BLOCK) void dist_advance_kernel(pipe_args a, bfs_policy_dist pol, int part,
                                                                 int parts) {
  __shared__ advance_smem<bfs_policy_dist> sm;
  ctrl_t* c = a.ctrl;
  if (c->done || c->mode != 0) return;
  pol.begin(c);
  advance_block<bfs_policy_dist, false>(a, c, pol, sm, c->level & 1, blockIdx.x * parts + part, gridDim.x * parts,
                                        c->total_chunks, a.chunk_tile);
}

// After the exchange.  Top-down: claim what the peers discovered in this rank's slice.
// Bottom-up: scan the in-edges of the unvisited owned vertices against the whole-graph
// frontier bitmap the all-to-all assembled in `recv`.
__global__ __launch_bounds__(ADV_BLOCK) void dist_post_kernel(pipe_args a, dobfs_args d, dist_args x) {
  __shared__ words_smem sm;
  __shared__ bottomup_smem<4, false> bsm;
  ctrl_t* c = a.ctrl;
  if (c->done) return;
  if (c->mode == 1) {
    bfs_bottomup_block(a, d, c, bsm);
    return;
  }
  const int depth = c->level + 1;
  const int q = depth & 1;
  words_to_tiles(a, c, q, x.slice_words, x.lo, [&](int w) {
    unsigned cand = 0u;
    for (int k = 0; k < x.parts; ++k)
      for (int j = 0; j < x.n_ranks; ++j)
        if (j != x.my_rank) cand |= x.recv[((size_t)k * x.n_ranks + j) * x.slice_words + w];
    unsigned acc = 0u;
    while (cand) {
      const int b = __ffs(cand) - 1;
      cand &= cand - 1;
      const int v = x.lo + w * 32 + b;  // each owned vertex is looked at by exactly one thread
      if (v < x.hi && d.dist[v] > depth) {
        d.dist[v] = depth;
        acc |= 1u << b;
      }
    }
    return acc;
  }, sm);
}

// Statistics of the frontier the next level expands (this rank's share): the input of the
// all_reduce that drives termination and the direction choice.  <<<1, 1024>>>
__global__ __launch_bounds__(PLAN_BLOCK) void dist_stats_kernel(pipe_args a, dobfs_args d, dist_args x) {
  __shared__ unsigned long long s_red[4];
  ctrl_t* c = a.ctrl;
  const int tid = threadIdx.x;
  if (tid < 4) s_red[tid] = 0ull;
  __syncthreads();
  if (c->done) {
    if (tid < 4) x.stats_local[tid] = 0;
    return;
  }
  const int q = (c->level + 1) & 1;
  long long n = 0, m = 0, op = 0, pr = 0;
  if (c->frontier_bitmap) {
    for (int i = tid; i < d.bu_grid; i += PLAN_BLOCK) {
      n += d.bu_part[4 * i];
      m += d.bu_part[4 * i + 1];
      op += d.bu_part[4 * i + 2];
      pr += d.bu_part[4 * i + 3];
    }
  } else {
    const int nt = c->n_tiles[q];
    for (int i = tid; i < nt; i += PLAN_BLOCK) {
      n += a.tile_count[i];
      m += a.tile_sums[i];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_xor(n, o, 64);
    m += __shfl_xor(m, o, 64);
    op += __shfl_xor(op, o, 64);
    pr += __shfl_xor(pr, o, 64);
  }
  if (dev::lane_id() == 0) {
    atomicAdd(&s_red[0], (unsigned long long)n);
    atomicAdd(&s_red[1], (unsigned long long)m);
    atomicAdd(&s_red[2], (unsigned long long)op);
    atomicAdd(&s_red[3], (unsigned long long)pr);
  }
  __syncthreads();
  if (tid == 0) {
    c->n_items[q] = (int)s_red[0];
    c->q_edges[q] = (long long)s_red[1];
    c->bu_open += (long long)s_red[2];
    c->bu_probes += (long long)s_red[3];
    x.stats_local[0] = (long long)s_red[0];
    x.stats_local[1] = (long long)s_red[1];
    x.stats_local[2] = 0;
    x.stats_local[3] = 0;
  }
}

template <class Kernel>
static int resident_per_cu(Kernel k) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, ADV_BLOCK, 0) != hipSuccess || n < 1) n = 4;
  return n > 8 ? 8 : n;
}

}  // namespace grx

using namespace grx;

// RCCL, resolved at run time: libgrx.so carries no link-time dependency on it (single-GPU users
// never load it), and the collectives of a level group can be issued from C -- one host call per
// group, capturable into a HIP graph without any Python in between.
struct rccl_api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
static rccl_api& rccl() {
  static rccl_api api = [] {
    rccl_api a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (!a.lib) return a;
    auto sym = [&](const char* n) { return dlsym(a.lib, n); };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv &&
           a.AllReduce && a.GetErrorString;
    return a;
  }();
  return api;
}
#define GRX_NCCL(expr)                                                                         \
  do {                                                                                         \
    ncclResult_t _r = (expr);                                                                  \
    if (_r != ncclSuccess)                                                                     \
      return ::grx::fail(GRX_ERROR_HIP, std::string("RCCL: ") + rccl().GetErrorString(_r) + "\t: " #expr); \
  } while (0)

struct grx_bfs_dist {
  ncclComm_t comm = nullptr;          // own communicator (grx_bfs_dist_comm_init), null: the caller runs the collectives
  hipGraphExec_t group_graph = nullptr;  // one captured level group (kernels + both collectives)
  int32_t* graph_labels = nullptr;    // label base pointer / direction the captured group was recorded for
  int32_t graph_dir = -1;
  bool graph_failed = false;
  grx_context_t ctx = nullptr;
  grx_graph_t g = nullptr;      // out-rows of the owned slice
  grx_graph_t g_in = nullptr;   // in-rows (null: symmetric, or direction optimisation off)
  pipe_args a{};
  dobfs_args d{};
  dist_args x{};
  int grid_advance = 0, grid_post = 0;
  bool active = false;
};

extern "C" {

int32_t grx_bfs_dist_slice_bits(int32_t n_vertices, int32_t n_ranks) {
  if (n_vertices < 0 || n_ranks < 1) return 0;
  const long long per = ((long long)n_vertices + n_ranks - 1) / n_ranks;
  const long long s = ((per + 2047) / 2048) * 2048;
  return (int32_t)(s < 2048 ? 2048 : s);
}

grx_status_t grx_bfs_dist_create(grx_context_t ctx, grx_graph_t out_rows, grx_graph_t in_rows, int32_t n_ranks,
                                 int32_t my_rank, long long n_edges_global, int32_t parts, void* d_send,
                                 void* d_recv, long long* d_stats_local, const long long* d_stats_global,
                                 grx_bfs_dist_t* out) {
  if (!ctx || !out_rows || !d_send || !d_recv || !d_stats_local || !d_stats_global || !out)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_create: null argument");
  if (n_ranks < 1 || n_ranks > DIST_MAX_RANKS || my_rank < 0 || my_rank >= n_ranks)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_create: bad rank layout");
  if (parts != 1 && parts != 2) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_create: parts must be 1 or 2");
  if (in_rows && in_rows->V != out_rows->V)
    return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_create: in-rows and out-rows disagree on V");
  GRX_HIP(hipSetDevice(ctx->device));
  grx_bfs_dist* h = new grx_bfs_dist();
  h->ctx = ctx;
  h->g = out_rows;
  h->g_in = in_rows;
  const int32_t S = grx_bfs_dist_slice_bits(out_rows->V, n_ranks);
  const long long lo = (long long)my_rank * S, hi = lo + S;
  dist_args& x = h->x;
  x.n_ranks = n_ranks;
  x.my_rank = my_rank;
  x.lo = (int32_t)(lo < out_rows->V ? lo : out_rows->V);
  x.hi = (int32_t)(hi < out_rows->V ? hi : out_rows->V);
  x.slice_words = S / 32;
  x.parts = parts;
  x.send = static_cast<unsigned*>(d_send);
  x.recv = static_cast<const unsigned*>(d_recv);
  x.stats_local = d_stats_local;
  x.stats_global = d_stats_global;
  x.e_global = n_edges_global;
  grx_status_t rc = pipeline_prepare(ctx, out_rows, &h->a);
  if (rc != GRX_SUCCESS) { delete h; return rc; }
  static int per_cu_adv = 0, per_cu_post = 0;
  if (!per_cu_adv) per_cu_adv = resident_per_cu(dist_advance_kernel);
  if (!per_cu_post) per_cu_post = resident_per_cu(dist_post_kernel);
  h->grid_advance = ctx->num_cus * per_cu_adv;
  h->grid_post = ctx->num_cus * per_cu_post;
  const size_t words = (size_t)n_ranks * x.slice_words;
  GRX_HIP(ctx->bitmap[0].reserve((size_t)x.slice_words * sizeof(unsigned)));          // visited slice
  GRX_HIP(ctx->bitmap[1].reserve((size_t)2 * x.slice_words * sizeof(unsigned)));      // frontier slices
  GRX_HIP(ctx->labels.reserve(words * sizeof(unsigned)));                             // `sent`
  GRX_HIP(ctx->bu_part.reserve((size_t)h->grid_post * 4 * sizeof(long long)));
  x.sent = ctx->labels.as<unsigned>();
  dobfs_args& d = h->d;
  const bool can_bottom_up = in_rows != nullptr || out_rows->symmetric;
  d.t_ro = in_rows ? in_rows->ro : out_rows->ro;
  d.t_ci = in_rows ? in_rows->ci : out_rows->ci;
  d.visited = ctx->bitmap[0].as<unsigned>();
  d.fbits[0] = ctx->bitmap[1].as<unsigned>();
  d.fbits[1] = d.fbits[0] + x.slice_words;
  d.n_words = x.slice_words;
  d.enabled = can_bottom_up ? 1 : 0;
  d.bu_part = ctx->bu_part.as<long long>();
  d.bu_grid = h->grid_post;
  d.ch_lo = x.lo / 64;
  d.fin_global = x.recv;
  *out = h;
  return GRX_SUCCESS;
}

// problem.reset() + frontier seed.  The caller all-reduces stats_local into stats_global
// before the first grx_bfs_dist_pre.  advance_direction: GRX_DIR_FORWARD keeps every level
// top-down; GRX_DIR_OPTIMIZED enables the bottom-up step (needs in-rows or symmetry).
// Sharded labels: d_local holds ONLY the owned slice (S = grx_bfs_dist_slice_bits entries, vertex v at
// d_local[v - rank * S]).  Every kernel of the partitioned enactor dereferences labels of owned vertices only,
// so they all work on the base pointer d_local - lo.
grx_status_t grx_bfs_dist_begin_local(grx_bfs_dist_t h, int32_t source, int32_t advance_direction, int32_t* d_local) {
  if (!h || !d_local) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin_local: null argument");
  return grx_bfs_dist_begin(h, source, advance_direction, d_local - h->x.lo);
}

grx_status_t grx_bfs_dist_begin(grx_bfs_dist_t h, int32_t source, int32_t advance_direction, int32_t* d_dist) {
  if (!h || !d_dist) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin: null argument");
  if (source < 0 || source >= h->g->V) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_begin: source out of range");
  grx_context_t ctx = h->ctx;
  GRX_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  dist_args& x = h->x;
  x.do_enabled = (advance_direction == GRX_DIR_OPTIMIZED && h->d.enabled) ? 1 : 0;
  h->d.dist = d_dist;
  if (x.hi > x.lo) GRX_HIP(fill_i32(s, d_dist + x.lo, INT_MAX, x.hi - x.lo));  // only the owned range is used
  GRX_HIP(hipMemsetAsync(x.sent, 0, (size_t)x.n_ranks * x.slice_words * sizeof(unsigned), s));
  ctx->h_mailbox[0] = 0;
  GRX_HIP(hipEventRecord(ctx->ev_begin, s));
  const int src_if_owned = (source >= x.lo && source < x.hi) ? source : -1;
  hipLaunchKernelGGL(dist_init_kernel, dim3(1), dim3(TILE), 0, s, h->a, x, d_dist, src_if_owned);
  GRX_HIP(hipGetLastError());
  h->active = true;
  return GRX_SUCCESS;
}

// Enqueue the part of a level group that precedes the exchange of half `part`:
//   part 0: head (termination, direction, chunk map) -> prep -> top-down advance of half 0
//   part 1: top-down advance of half 1 (only with parts == 2)
// After part k the caller exchanges send[k] -> recv[k] (all_to_all_single, n_ranks equal
// splits of slice_words words).  Asynchronous.
grx_status_t grx_bfs_dist_pre(grx_bfs_dist_t h, int32_t part) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_pre: no BFS in flight");
  if (part < 0 || part >= h->x.parts) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_pre: bad part");
  grx_context_t ctx = h->ctx;
  hipStream_t s = ctx->stream;
  const dist_args& x = h->x;
  if (part == 0) {
    hipLaunchKernelGGL(dist_head_kernel, dim3(1), dim3(PLAN_BLOCK), 0, s, h->a, x);
    hipLaunchKernelGGL(dist_prep_kernel, dim3(ctx->num_cus * 2), dim3(ADV_BLOCK), 0, s, h->a, h->d, x);
  }
  bfs_policy_dist pol{h->d.dist, x.sent, x.send + (size_t)part * x.n_ranks * x.slice_words, x.lo, x.hi, 0};
  hipLaunchKernelGGL(dist_advance_kernel, dim3(h->grid_advance), dim3(ADV_BLOCK), 0, s, h->a, pol, part, x.parts);
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

// Enqueue the part of a level group that follows the exchange(s): apply / bottom-up, then
// the statistics kernel.  The caller then all-reduces stats_local into stats_global.
grx_status_t grx_bfs_dist_post(grx_bfs_dist_t h) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_post: no BFS in flight");
  grx_context_t ctx = h->ctx;
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(dist_post_kernel, dim3(h->grid_post), dim3(ADV_BLOCK), 0, s, h->a, h->d, h->x);
  hipLaunchKernelGGL(dist_stats_kernel, dim3(1), dim3(PLAN_BLOCK), 0, s, h->a, h->d, h->x);
  GRX_HIP(hipGetLastError());
  return GRX_SUCCESS;
}

// Wait for everything enqueued so far and report the state of the search.
grx_status_t grx_bfs_dist_poll(grx_bfs_dist_t h, int32_t* done, int32_t* level) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_poll: no BFS in flight");
  grx_context_t ctx = h->ctx;
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, ctx->stream));
  GRX_HIP(hipStreamSynchronize(ctx->stream));
  if (done) *done = ctx->h_ctrl->done;
  if (level) *level = ctx->h_ctrl->level;
  return GRX_SUCCESS;
}

grx_status_t grx_bfs_dist_end(grx_bfs_dist_t h, grx_run_stats_t* stats) {
  if (!h || !h->active) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_end: no BFS in flight");
  grx_context_t ctx = h->ctx;
  hipStream_t s = ctx->stream;
  GRX_HIP(hipEventRecord(ctx->ev_end, s));
  GRX_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->d_ctrl, sizeof(ctrl_t), hipMemcpyDeviceToHost, s));
  GRX_HIP(hipEventSynchronize(ctx->ev_end));
  GRX_HIP(hipStreamSynchronize(s));
  float ms = 0;
  GRX_HIP(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
  ctx->stats.edges_visited = ctx->h_ctrl->edges_visited;
  ctx->stats.vertices_visited = ctx->h_ctrl->vertices_visited;
  ctx->stats.search_depth = ctx->h_ctrl->level;
  ctx->stats.elapsed_ms = ms;
  ctx->stats.n_levels_recorded = 0;
  if (stats) *stats = ctx->stats;
  h->active = false;
  return GRX_SUCCESS;
}

// ---- RCCL transport inside the library --------------------------------------------------------
int32_t grx_dist_unique_id_bytes(void) { return (int32_t)sizeof(ncclUniqueId); }

grx_status_t grx_dist_unique_id(void* out) {
  if (!out) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_dist_unique_id: null argument");
  if (!rccl().ok) return fail(GRX_ERROR_UNSUPPORTED, "grx_dist_unique_id: librccl could not be opened");
  GRX_NCCL(rccl().GetUniqueId(reinterpret_cast<ncclUniqueId*>(out)));
  return GRX_SUCCESS;
}

grx_status_t grx_bfs_dist_comm_init(grx_bfs_dist_t h, const void* unique_id) {
  if (!h || !unique_id) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_comm_init: null argument");
  if (!rccl().ok) return fail(GRX_ERROR_UNSUPPORTED, "grx_bfs_dist_comm_init: librccl could not be opened");
  if (h->x.parts != 1) return fail(GRX_ERROR_UNSUPPORTED, "grx_bfs_dist_comm_init: the in-library transport runs one exchange per level (parts == 1)");
  GRX_HIP(hipSetDevice(h->ctx->device));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  GRX_NCCL(rccl().CommInitRank(&h->comm, h->x.n_ranks, id, h->x.my_rank));
  return GRX_SUCCESS;
}

// one level group, everything on the context's stream: kernels, the bitmap all-to-all (grouped
// send / recv: every pair of GPUs has its own xGMI link) and the 4-word statistics all-reduce
static grx_status_t dist_group_enqueue(grx_bfs_dist_t h) {
  grx_status_t st = grx_bfs_dist_pre(h, 0);
  if (st != GRX_SUCCESS) return st;
  const dist_args& x = h->x;
  hipStream_t s = h->ctx->stream;
  GRX_NCCL(rccl().GroupStart());
  for (int j = 0; j < x.n_ranks; ++j) {
    GRX_NCCL(rccl().Send(x.send + (size_t)j * x.slice_words, (size_t)x.slice_words, ncclUint32, j, h->comm, s));
    GRX_NCCL(rccl().Recv(const_cast<unsigned*>(x.recv) + (size_t)j * x.slice_words, (size_t)x.slice_words, ncclUint32, j,
                         h->comm, s));
  }
  GRX_NCCL(rccl().GroupEnd());
  st = grx_bfs_dist_post(h);
  if (st != GRX_SUCCESS) return st;
  GRX_NCCL(rccl().AllReduce(x.stats_local, const_cast<long long*>(x.stats_global), 4, ncclInt64, ncclSum, h->comm, s));
  return GRX_SUCCESS;
}

// the all-reduce that follows grx_bfs_dist_begin (frontier statistics of the seed)
grx_status_t grx_bfs_dist_seed_stats(grx_bfs_dist_t h) {
  if (!h || !h->comm) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_seed_stats: no communicator");
  GRX_NCCL(rccl().AllReduce(h->x.stats_local, const_cast<long long*>(h->x.stats_global), 4, ncclInt64, ncclSum, h->comm,
                            h->ctx->stream));
  return GRX_SUCCESS;
}

// Enqueue n level groups.  A group takes no level-dependent argument, so once a search has finished the
// group is captured into a HIP graph (grx_bfs_dist_capture_group) and later calls replay it: one graph
// launch per level; without a captured graph the groups are enqueued eagerly.
grx_status_t grx_bfs_dist_groups(grx_bfs_dist_t h, int32_t n) {
  if (!h || !h->active || !h->comm) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_groups: no BFS in flight or no communicator");
  const bool replay = h->group_graph && h->graph_labels == h->d.dist && h->graph_dir == h->x.do_enabled;
  for (int i = 0; i < n; ++i) {
    if (replay) {
      GRX_HIP(hipGraphLaunch(h->group_graph, h->ctx->stream));
    } else {
      grx_status_t st = dist_group_enqueue(h);
      if (st != GRX_SUCCESS) return st;
    }
  }
  return GRX_SUCCESS;
}

// Record one level group for the label buffer / direction of the search that just ended (every kernel of a
// further group exits on `done`, and capture only records).  Must be called on every rank alike.  A failure
// is remembered and leaves the eager path in place.
grx_status_t grx_bfs_dist_capture_group(grx_bfs_dist_t h) {
  if (!h || !h->comm) return fail(GRX_ERROR_INVALID_ARGUMENT, "grx_bfs_dist_capture_group: no communicator");
  if (h->graph_failed) return GRX_SUCCESS;
  if (h->group_graph && h->graph_labels == h->d.dist && h->graph_dir == h->x.do_enabled) return GRX_SUCCESS;
  if (h->group_graph) { (void)hipGraphExecDestroy(h->group_graph); h->group_graph = nullptr; }
  hipStream_t s = h->ctx->stream;
  hipGraph_t graph = nullptr;
  const bool was_active = h->active;
  h->active = true;
  bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
  if (ok) {
    ok = dist_group_enqueue(h) == GRX_SUCCESS;
    ok = (hipStreamEndCapture(s, &graph) == hipSuccess) && ok && graph != nullptr;
    if (ok) ok = hipGraphInstantiate(&h->group_graph, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
  }
  (void)hipGetLastError();
  h->active = was_active;
  if (!ok) {
    h->group_graph = nullptr;
    h->graph_failed = true;
    return GRX_SUCCESS;
  }
  h->graph_labels = h->d.dist;
  h->graph_dir = h->x.do_enabled;
  return GRX_SUCCESS;
}

int32_t grx_bfs_dist_group_is_captured(grx_bfs_dist_t h) { return (h && h->group_graph) ? 1 : 0; }

grx_status_t grx_bfs_dist_destroy(grx_bfs_dist_t h) {
  if (h) {
    if (h->group_graph) (void)hipGraphExecDestroy(h->group_graph);
    if (h->comm && rccl().ok) (void)rccl().CommDestroy(h->comm);
  }
  delete h;
  return GRX_SUCCESS;
}

}  // extern "C"
