// filter.hxx -- the filter operator: keep the valid elements of the input
// frontier for which `op(v)` holds.
// API parity: include/gunrock/framework/operators/filter/filter.hxx:72-211
// (reference): execute<alg>(G, op, in*, out*, context); enactor overload
// execute<alg>(G, E, op, context, swap_buffers = true); execute_runtime(G, E, op,
// alg, context, swap).  `op` is never called on an invalid (-1) element and is
// called exactly once per valid element (SSSP's stamp filter has side effects).
// Algorithms:
//   predicated / remove  stable compaction (the reference: rocThrust copy_if /
//                        remove_copy_if, filter/predicated.hxx:24-39, remove.hxx:23-40)
//                        -> flag + block counts, scan of block counts, scatter.
//   compact              single pass: wave ballot + mbcnt prefix inside a wave, LDS
//                        wave totals, one atomicAdd per workgroup for its output
//                        base.  THROWS in the reference (filter/compact.hxx:21-24).
//                        Not stable across workgroups.
//   bypass               no compaction, out[i] = keep ? in[i] : -1 (in place allowed)
//                        (filter/bypass.hxx:31-69).
#pragma once

#include <gunrock/util/trace.hxx>

#include <gunrock/cuda/context.hxx>
#include <gunrock/error.hxx>
#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/hip/scan.hxx>
#include <gunrock/hip/wave.hxx>
#include <gunrock/util/type_limits.hxx>

namespace gunrock {
namespace operators {
namespace filter {
namespace detail {

constexpr int BLOCK = 256;
constexpr int ITEMS = 8;
constexpr int TILE = BLOCK * ITEMS;

template <typename operator_t, typename type_t>
__global__ __launch_bounds__(BLOCK) void bypass_kernel(operator_t op, const type_t* in, type_t* out, std::size_t n) {
  for (std::size_t i = (std::size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (std::size_t)gridDim.x * BLOCK) {
    const type_t v = in[i];
    const bool keep = gunrock::util::limits::is_valid(v) && op(v);
    out[i] = keep ? v : gunrock::numeric_limits<type_t>::invalid();
  }
}

// pass 1 of the stable compaction: evaluate the predicate once, keep the flags
template <typename operator_t, typename type_t>
__global__ __launch_bounds__(BLOCK) void flag_kernel(operator_t op, const type_t* in, std::size_t n,
                                                     unsigned char* flags, int32_t* tile_counts) {
  __shared__ int s_w[BLOCK / 64];
  const std::size_t base = (std::size_t)blockIdx.x * TILE;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const std::size_t i = base + (std::size_t)k * BLOCK + threadIdx.x;
    if (i < n) {
      const type_t v = in[i];
      const bool keep = gunrock::util::limits::is_valid(v) && op(v);
      flags[i] = keep ? 1 : 0;
      cnt += keep ? 1 : 0;
    }
  }
  cnt = grx::dev::wave_sum(cnt);
  if (grx::dev::lane_id() == 0) s_w[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) t += s_w[i];
    tile_counts[blockIdx.x] = t;
  }
}

// pass 2: order-preserving scatter; element i of a tile precedes element j > i
template <typename type_t>
__global__ __launch_bounds__(BLOCK) void scatter_kernel(const type_t* in, std::size_t n, const unsigned char* flags,
                                                        const int32_t* tile_offsets, type_t* out) {
  __shared__ int s_w[BLOCK / 64 + 1];
  const std::size_t base = (std::size_t)blockIdx.x * TILE;
  int running = tile_offsets[blockIdx.x];
  const int lane = grx::dev::lane_id();
  const int wid = threadIdx.x >> 6;
#pragma unroll 1
  for (int k = 0; k < ITEMS; ++k) {
    const std::size_t i = base + (std::size_t)k * BLOCK + threadIdx.x;
    const bool keep = i < n && flags[i];
    const unsigned long long m = grx::dev::ballot(keep);
    if (lane == 0) s_w[wid] = __popcll(m);
    __syncthreads();
    int before = 0, row = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
      const int c = s_w[w];
      if (w < wid) before += c;
      row += c;
    }
    if (keep) out[running + before + grx::dev::mask_rank(m)] = in[i];
    running += row;
    __syncthreads();
  }
}

// single-pass compaction (unordered across workgroups)
template <typename operator_t, typename type_t>
__global__ __launch_bounds__(BLOCK) void compact_kernel(operator_t op, const type_t* in, std::size_t n, type_t* out,
                                                        int32_t* counter) {
  __shared__ int s_w[BLOCK / 64 + 1];
  __shared__ int s_base;
  const int lane = grx::dev::lane_id();
  const int wid = threadIdx.x >> 6;
  for (std::size_t base = (std::size_t)blockIdx.x * TILE; base < n; base += (std::size_t)gridDim.x * TILE) {
    type_t kept[ITEMS];
    int rank[ITEMS];
    int mine = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const std::size_t i = base + (std::size_t)k * BLOCK + threadIdx.x;
      type_t v = gunrock::numeric_limits<type_t>::invalid();
      bool keep = false;
      if (i < n) {
        v = in[i];
        keep = gunrock::util::limits::is_valid(v) && op(v);
      }
      const unsigned long long m = grx::dev::ballot(keep);
      kept[k] = v;
      rank[k] = keep ? mine + grx::dev::mask_rank(m) : -1;  // rank inside this wave so far
      mine += __popcll(m);                                   // wave-uniform
    }
    if (lane == 0) s_w[wid] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
#pragma unroll
      for (int w = 0; w < BLOCK / 64; ++w) tot += s_w[w];
      s_base = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wid; ++w) before += s_w[w];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
      if (rank[k] >= 0) out[before + rank[k]] = kept[k];
    __syncthreads();
  }
}

// SMALL inputs (round 5): predicate, scan, ordered scatter and the count in ONE launch of one workgroup; the count goes
// straight to the host's mailbox.  Stable, so it serves every compacting algorithm (predicated / remove: flag + three-launch
// scan + scatter + copy were six launches; compact: memset + kernel + copy).  <<<1, SMALL_BLOCK>>>, n <= SMALL_N
constexpr int SMALL_BLOCK = 1024, SMALL_PER = 8, SMALL_N = SMALL_BLOCK * SMALL_PER;
template <typename operator_t, typename type_t>
__global__ __launch_bounds__(SMALL_BLOCK) void compact_small_kernel(operator_t op, const type_t* in, int n, type_t* out,
                                                                    int* host_count) {
  __shared__ int s_w[SMALL_BLOCK / 64 + 1];
  type_t v[SMALL_PER];
  bool keep[SMALL_PER];
  int local = 0;
  const int first = (int)threadIdx.x * SMALL_PER;  // a thread owns consecutive elements: the output keeps the input's order
#pragma unroll
  for (int k = 0; k < SMALL_PER; ++k) {
    const int i = first + k;
    v[k] = gunrock::numeric_limits<type_t>::invalid();
    keep[k] = false;
    if (i < n) {
      v[k] = in[i];
      keep[k] = gunrock::util::limits::is_valid(v[k]) && op(v[k]);
    }
    local += keep[k] ? 1 : 0;
  }
  int tot;
  int at = grx::dev::block_exclusive_sum<SMALL_BLOCK>(local, s_w, &tot);
#pragma unroll
  for (int k = 0; k < SMALL_PER; ++k)
    if (keep[k]) out[at++] = v[k];
  if (threadIdx.x == 0) *host_count = tot;
}
// returns false when the small path does not apply (the caller takes the general one)
template <typename operator_t, typename type_t>
bool compact_small(operator_t op, const type_t* in, std::size_t n, type_t* out, gcuda::standard_context_t& ctx, std::size_t* kept) {
  if (n > (std::size_t)SMALL_N) return false;
  int* host_count = ctx.mailbox_device(1);
  if (!host_count) return false;
  hipLaunchKernelGGL((compact_small_kernel<operator_t, type_t>), dim3(1), dim3(SMALL_BLOCK), 0, ctx.stream(), op, in, (int)n, out,
                     host_count);
  *kept = (std::size_t)ctx.wait_mailbox(1)[0];
  return true;
}

inline unsigned strided_grid(std::size_t n, int per_block, gcuda::standard_context_t& ctx) {
  std::size_t g = (n + (std::size_t)per_block - 1) / (std::size_t)per_block;
  const std::size_t cap = (std::size_t)ctx.props().multiProcessorCount * 8;
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

// Generic stable compaction of in[0..n) by a predicate evaluated once per
// valid element; returns the number kept (host value, one stream sync).
template <typename operator_t, typename type_t>
std::size_t stable_compact(operator_t op, const type_t* in, std::size_t n, type_t* out,
                           gcuda::standard_context_t& ctx) {
  if (n == 0) return 0;
  std::size_t kept_small;
  if (compact_small(op, in, n, out, ctx, &kept_small)) return kept_small;
  const std::size_t tiles = (n + TILE - 1) / TILE;
  unsigned char* flags = ctx.scratch<unsigned char>(1, n);
  int32_t* counts = ctx.scratch<int32_t>(2, tiles + 2);
  int32_t* sums = ctx.scratch<int32_t>(0, (std::size_t)grx::scan_num_blocks((int64_t)tiles) + 2);
  hipStream_t s = ctx.stream();
  hipLaunchKernelGGL((flag_kernel<operator_t, type_t>), dim3((unsigned)tiles), dim3(BLOCK), 0, s, op, in, n, flags,
                     counts);
  grx::exclusive_scan_i32(s, counts, (int64_t)tiles, counts, sums);
  hipLaunchKernelGGL((scatter_kernel<type_t>), dim3((unsigned)tiles), dim3(BLOCK), 0, s, in, n, flags, counts, out);
  return (std::size_t)ctx.read_back(counts + tiles)[0];
}

}  // namespace detail

// Per-algorithm entry points with the reference's names and signatures (filter/predicated.hxx:12-39, remove.hxx:11-44,
// bypass.hxx:13-69, compact.hxx:13-25): filter::<algorithm>::execute(G, op, input, output, standard_context).  The
// headers of those names forward here.
namespace detail {
template <typename frontier_t>
bool prepare_output(frontier_t* input, frontier_t* output) {
  const std::size_t n = input->get_number_of_elements();
  if (output != input && output->get_capacity() < n) output->reserve(n);
  if (n == 0) output->set_number_of_elements(0);
  return n != 0;
}
}  // namespace detail

namespace predicated {
template <typename graph_t, typename operator_t, typename frontier_t>
void execute(graph_t& G, operator_t op, frontier_t* input, frontier_t* output, gcuda::standard_context_t& context) {
  (void)G;
  if (!detail::prepare_output(input, output)) return;
  error::throw_if_exception(output == input, "stable filter cannot run in place");
  output->set_number_of_elements(
      detail::stable_compact(op, input->data(), input->get_number_of_elements(), output->data(), context));
}
}  // namespace predicated

namespace remove {  // identical result: the same stable compaction
template <typename graph_t, typename operator_t, typename frontier_t>
void execute(graph_t& G, operator_t op, frontier_t* input, frontier_t* output, gcuda::standard_context_t& context) {
  predicated::execute(G, op, input, output, context);
}
}  // namespace remove

namespace compact {  // throws in the reference (filter/compact.hxx:21-24); real here
template <typename graph_t, typename operator_t, typename frontier_t>
void execute(graph_t& G, operator_t op, frontier_t* input, frontier_t* output, gcuda::standard_context_t& context) {
  (void)G;
  using type_t = typename frontier_t::type_t;
  if (!detail::prepare_output(input, output)) return;
  error::throw_if_exception(output == input, "compact filter cannot run in place");
  const std::size_t n = input->get_number_of_elements();
  std::size_t kept_small;
  if (detail::compact_small(op, input->data(), n, output->data(), context, &kept_small)) {
    output->set_number_of_elements(kept_small);
    return;
  }
  int32_t* counter = context.template scratch<int32_t>(2, 4);
  error::throw_if_exception(hipMemsetAsync(counter, 0, sizeof(int32_t), context.stream()), "counter reset");
  hipLaunchKernelGGL((detail::compact_kernel<operator_t, type_t>), dim3(detail::strided_grid(n, detail::TILE, context)),
                     dim3(detail::BLOCK), 0, context.stream(), op, input->data(), n, output->data(), counter);
  output->set_number_of_elements((std::size_t)context.read_back(counter)[0]);
}
}  // namespace compact

namespace bypass {
template <typename graph_t, typename operator_t, typename frontier_t>
void execute(graph_t& G, operator_t op, frontier_t* input, frontier_t* output, gcuda::standard_context_t& context) {
  (void)G;
  using type_t = typename frontier_t::type_t;
  if (!detail::prepare_output(input, output)) return;
  const std::size_t n = input->get_number_of_elements();
  hipLaunchKernelGGL((detail::bypass_kernel<operator_t, type_t>), dim3(detail::strided_grid(n, detail::BLOCK, context)),
                     dim3(detail::BLOCK), 0, context.stream(), op, input->data(), output->data(), n);
  output->set_number_of_elements(n);
}
template <typename graph_t, typename operator_t, typename frontier_t>
void execute(graph_t& G, operator_t op, frontier_t* input, gcuda::standard_context_t& context) {  // in place
  execute(G, op, input, input, context);
}
}  // namespace bypass

template <filter_algorithm_t alg_type, typename graph_t, typename operator_t, typename frontier_t>
void execute(graph_t& G, operator_t op, frontier_t* input, frontier_t* output, gcuda::multi_context_t& context) {
  GUNROCK_TRACE_RANGE("filter");
  error::throw_if_exception(context.size() != 1, "`context.size() != 1` not supported");
  auto& ctx = *context.get_context(0);
  if constexpr (alg_type == filter_algorithm_t::bypass) bypass::execute(G, op, input, output, ctx);
  else if constexpr (alg_type == filter_algorithm_t::compact) compact::execute(G, op, input, output, ctx);
  else if constexpr (alg_type == filter_algorithm_t::remove) remove::execute(G, op, input, output, ctx);
  else predicated::execute(G, op, input, output, ctx);
}

// in-place bypass overload (filter/bypass.hxx:62-69 of the reference)
template <filter_algorithm_t alg_type, typename graph_t, typename operator_t, typename frontier_t>
void execute(graph_t& G, operator_t op, frontier_t* input, gcuda::multi_context_t& context) {
  static_assert(alg_type == filter_algorithm_t::bypass, "only bypass can filter in place");
  execute<alg_type>(G, op, input, input, context);
}

template <filter_algorithm_t alg_type, typename graph_t, typename enactor_type, typename operator_t>
void execute(graph_t& G, enactor_type* E, operator_t op, gcuda::multi_context_t& context,
             bool swap_buffers = true) {
  execute<alg_type>(G, op, E->get_input_frontier(), E->get_output_frontier(), context);
  if (swap_buffers) E->swap_frontier_buffers();
}

template <typename graph_t, typename enactor_type, typename operator_t>
void execute_runtime(graph_t& G, enactor_type* E, operator_t op, filter_algorithm_t alg_type,
                     gcuda::multi_context_t& context, bool swap_buffers = true) {
  switch (alg_type) {
    case filter_algorithm_t::remove:
      execute<filter_algorithm_t::remove>(G, E, op, context, swap_buffers);
      break;
    case filter_algorithm_t::predicated:
      execute<filter_algorithm_t::predicated>(G, E, op, context, swap_buffers);
      break;
    case filter_algorithm_t::compact:
      execute<filter_algorithm_t::compact>(G, E, op, context, swap_buffers);
      break;
    case filter_algorithm_t::bypass:
      execute<filter_algorithm_t::bypass>(G, E, op, context, swap_buffers);
      break;
    default:
      error::throw_if_exception(hipErrorUnknown, "Filter algorithm type not supported.");
  }
}

}  // namespace filter
}  // namespace operators
}  // namespace gunrock
