// helpers.hxx -- output sizing and the shared LDS expansion of advance.
// API parity: include/gunrock/framework/operators/advance/helpers.hxx:41-161
// (reference): compute_output_offsets (exclusive scan of the degrees of the
// input frontier into `segments[0..n]`, invalid slots count 0) and
// compute_output_length.  The reference runs rocThrust transform_exclusive_scan
// and copies the total back through a 1-element host_vector; here the degree
// transform is fused into our own three-launch scan and the total comes back
// through the context's pinned mailbox (one stream sync per advance).
#pragma once

#include <hip/hip_runtime.h>

#include <utility>

#include <gunrock/cuda/context.hxx>
#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/hip/scan.hxx>
#include <gunrock/hip/wave.hxx>
#include <gunrock/memory.hxx>
#include <gunrock/util/type_limits.hxx>

namespace gunrock {
namespace operators {
namespace advance {
namespace detail {

constexpr int BLOCK = 256;                // threads per workgroup == slots per window
constexpr int ATOMS_PER_BLOCK = 2048;     // merge-path: edges per workgroup

// degrees of the input slots (invalid slots: 0) and, in *max_out (zeroed by the caller), the longest row among them
template <typename graph_t, typename type_t, typename edge_t>
__global__ void degrees_kernel(graph_t G, const type_t* input, std::size_t n, edge_t* degrees, int* max_out) {
  int mx = 0;
  for (std::size_t i = (std::size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * blockDim.x) {
    const type_t v = input ? input[i] : (type_t)i;
    const edge_t d = gunrock::util::limits::is_valid(v) ? G.get_number_of_neighbors(v) : 0;
    degrees[i] = d;
    mx = max(mx, (int)d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
  if (max_out && (threadIdx.x & 63) == 0 && mx > 0) atomicMax(max_out, mx);
}
// {total, longest row} of a frontier -> the host's mailbox (one launch instead of a device-to-host copy)
template <typename edge_t>
__global__ void publish_total_kernel(const edge_t* total, const int* max_in, int* host_words) {
  host_words[0] = (int)*total;
  host_words[1] = *max_in;
}

// SMALL frontiers (round 5): degrees + scan + total in ONE launch of one workgroup, the total written where the host reads it.
// The general path is four launches (degrees, three-launch scan) and a device-to-host copy; on an algorithm of thousands of
// short iterations (the reference's k-core on these operators: ~2300 advances of a few hundred vertices) that overhead was
// most of the run.  <<<1, SMALL_BLOCK>>>, n <= SMALL_N
constexpr int SMALL_BLOCK = 1024, SMALL_PER = 8, SMALL_N = SMALL_BLOCK * SMALL_PER;
template <typename graph_t, typename type_t, typename edge_t>
__global__ __launch_bounds__(SMALL_BLOCK) void degrees_scan_small_kernel(graph_t G, const type_t* input, int n, edge_t* segments,
                                                                         int* host_total) {
  __shared__ int s_w[SMALL_BLOCK / 64 + 1];
  int deg[SMALL_PER], local = 0;
  const int first = (int)threadIdx.x * SMALL_PER;
#pragma unroll
  for (int k = 0; k < SMALL_PER; ++k) {
    const int i = first + k;
    deg[k] = 0;
    if (i < n) {
      const type_t v = input ? input[i] : (type_t)i;
      if (gunrock::util::limits::is_valid(v)) deg[k] = (int)G.get_number_of_neighbors(v);
    }
    local += deg[k];
  }
  int tot;
  int ex = grx::dev::block_exclusive_sum<SMALL_BLOCK>(local, s_w, &tot);
  int mx = 0;
#pragma unroll
  for (int k = 0; k < SMALL_PER; ++k) {
    if (first + k < n) segments[first + k] = (edge_t)ex;
    ex += deg[k];
    mx = max(mx, deg[k]);
  }
  __shared__ int s_max;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(&s_max, mx);
  __syncthreads();
  if (threadIdx.x == 0) {
    segments[n] = (edge_t)tot;
    host_total[0] = tot;
    host_total[1] = s_max;  // the longest row of the frontier (advance::execute: hub-only frontiers go to the merge-path kernel)
  }
}

// Expand the edges ("atoms") [atom_lo, atom_hi) of a staged window of up to
// BLOCK input slots.  s_seg[0..nslots] is the window-relative exclusive degree
// scan, s_start the first edge id of each slot, s_src the slot's vertex.  Lanes
// take consecutive atoms, so column-index / weight reads are coalesced inside a
// row.  `out` points at the output position of window atom 0 (may be null).
//
// A thread works on ROUND atoms at a time, phase by phase: the ROUND owner searches step together (ROUND independent LDS
// probes in flight per step instead of one dependent chain per atom), then all row starts, then all neighbour ids /
// weights from clamped indices (atoms past the end read the window's first edge), then the user operator, then the stores.
// Written as one loop over atoms -- search, load, call, store -- every global load sat behind its own search and in
// front of its own consumer: one round trip at a time per lane (seen in the ISA of round 2's kernels).
constexpr int ROUND = 8;

// COMPACT (fused advance + compact filter, advance::execute_compact): instead of out[atom] = keep ? nbr : -1 the kept
// neighbours of a round leave compacted -- wave ballot + mbcnt rank, wave totals in LDS (s_cmp: BLOCK / 64 + 2 ints), ONE
// atomicAdd on *counter per workgroup and round for the output base -- so the -1 holes never reach HBM and the separate
// filter pass (read m_F, write the survivors) disappears.  Unordered across workgroups, like filter_algorithm_t::compact.
template <advance_io_type_t output_type, bool COMPACT = false, typename graph_t, typename operator_t, typename type_t>
__device__ __forceinline__ void expand_window(const graph_t& G, operator_t& op, const int* s_seg,
                                              const int* s_start, const type_t* s_src, int nslots,
                                              int atom_lo, int atom_hi, type_t* out, int* s_cmp = nullptr,
                                              int32_t* counter = nullptr) {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;
  for (int base = atom_lo; base < atom_hi; base += BLOCK * ROUND) {  // uniform over the workgroup
    int lo[ROUND], atom[ROUND];
#pragma unroll
    for (int k = 0; k < ROUND; ++k) {
      atom[k] = base + k * BLOCK + (int)threadIdx.x;
      lo[k] = 0;
    }
    // largest slot with s_seg[slot] <= atom, for all ROUND atoms at once (atoms past the end search for atom_hi - 1).
    // The BLOCK * ROUND atoms of a round are consecutive, so their owners lie between the owner of the first and the
    // owner of the last one: two uniform searches (broadcast reads) bound the range, and the per-atom search needs
    // log2(range) steps instead of 8 -- three on a frontier of hubs (hundreds of edges per row).
    int s_lo = 0, s_hi = 0;
    {
      const int a_first = base, a_last = min(base + BLOCK * ROUND, atom_hi) - 1;
#pragma unroll
      for (int step = BLOCK / 2; step >= 1; step >>= 1) {
        if (s_lo + step < nslots && s_seg[s_lo + step] <= a_first) s_lo += step;
        if (s_hi + step < nslots && s_seg[s_hi + step] <= a_last) s_hi += step;
      }
    }
    int top = 1;
    while (top * 2 <= s_hi - s_lo) top *= 2;  // uniform: largest power of two <= the range (1 when the range is 0 or 1)
#pragma unroll
    for (int k = 0; k < ROUND; ++k) lo[k] = s_lo;
    for (int step = top; step >= 1; step >>= 1) {
      int probe[ROUND];
#pragma unroll
      for (int k = 0; k < ROUND; ++k) probe[k] = s_seg[min(lo[k] + step, nslots)];
#pragma unroll
      for (int k = 0; k < ROUND; ++k)
        if (lo[k] + step <= s_hi && probe[k] <= min(atom[k], atom_hi - 1)) lo[k] += step;
    }
    edge_t e[ROUND];
    vertex_t src[ROUND];
#pragma unroll
    for (int k = 0; k < ROUND; ++k) {
      const int a = min(atom[k], atom_hi - 1);  // a real atom of the window: its edge exists
      e[k] = (edge_t)(s_start[lo[k]] + (a - s_seg[lo[k]]));
      src[k] = (vertex_t)s_src[lo[k]];
    }
    vertex_t nbr[ROUND];
    weight_t w[ROUND];
#pragma unroll
    for (int k = 0; k < ROUND; ++k) nbr[k] = G.get_destination_vertex(e[k]);
#pragma unroll
    for (int k = 0; k < ROUND; ++k) w[k] = G.get_edge_weight(e[k]);
    if constexpr (!COMPACT) {
#pragma unroll
      for (int k = 0; k < ROUND; ++k) {
        if (atom[k] < atom_hi) {
          // mutable lvalues: user operators may take (vertex_t&, vertex_t&, edge_t const&, weight_t const&)
          // like the reference's hits.hxx:137
          edge_t ee = e[k];
          vertex_t ss = src[k], nn = nbr[k];
          weight_t ww = w[k];
          const bool keep = op(ss, nn, ee, ww);
          if constexpr (output_type != advance_io_type_t::none) {
            const type_t emitted = (output_type == advance_io_type_t::edges) ? (type_t)e[k] : (type_t)nbr[k];
            out[atom[k]] = keep ? emitted : gunrock::numeric_limits<type_t>::invalid();
          }
        }
      }
    } else {
      static_assert(output_type != advance_io_type_t::none, "a compacted advance has an output");
      const int lane = grx::dev::lane_id();
      const int wid = (int)threadIdx.x >> 6;
      int rank[ROUND];
      int mine = 0;  // kept by this wave so far (wave-uniform)
#pragma unroll
      for (int k = 0; k < ROUND; ++k) {
        bool keep = false;
        if (atom[k] < atom_hi) {
          edge_t ee = e[k];
          vertex_t ss = src[k], nn = nbr[k];
          weight_t ww = w[k];
          keep = op(ss, nn, ee, ww);
        }
        const unsigned long long m = grx::dev::ballot(keep);
        rank[k] = keep ? mine + grx::dev::mask_rank(m) : -1;
        mine += __popcll(m);
      }
      if (lane == 0) s_cmp[wid] = mine;
      __syncthreads();
      if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int i = 0; i < BLOCK / 64; ++i) tot += s_cmp[i];
        s_cmp[BLOCK / 64] = tot ? atomicAdd(counter, tot) : 0;
      }
      __syncthreads();
      int before = s_cmp[BLOCK / 64];
      for (int i = 0; i < wid; ++i) before += s_cmp[i];
#pragma unroll
      for (int k = 0; k < ROUND; ++k)
        if (rank[k] >= 0) out[before + rank[k]] = (output_type == advance_io_type_t::edges) ? (type_t)e[k] : (type_t)nbr[k];
      __syncthreads();  // s_cmp is reused by the next round
    }
  }
}

}  // namespace detail

// segments[0..n] = exclusive scan of the input frontier's degrees; returns the
// total (host value).  input == nullptr means "every vertex of the graph".
// max_degree (optional): the longest row among the input's vertices
template <typename graph_t, typename type_t, typename work_tiles_t>
std::size_t compute_output_offsets(graph_t& G, const type_t* input, std::size_t n, work_tiles_t& segments,
                                   gcuda::standard_context_t& context, int* max_degree = nullptr) {
  using edge_t = typename graph_t::edge_type;
  static_assert(sizeof(edge_t) == 4, "offsets are 32-bit in this engine");
  if (segments.size() < n + 1) segments.resize(n + 1);
  edge_t* seg = memory::raw_pointer_cast(segments.data());
  if (max_degree) *max_degree = 0;
  if (n == 0) return 0;
  hipStream_t s = context.stream();
  int* host_words = context.mailbox_device(4);  // two words: {total, longest row}
  if (n <= (std::size_t)detail::SMALL_N && host_words) {
    hipLaunchKernelGGL((detail::degrees_scan_small_kernel<graph_t, type_t, edge_t>), dim3(1), dim3(detail::SMALL_BLOCK), 0, s, G,
                       input, (int)n, seg, host_words);
    const int* w = context.wait_mailbox(4);
    if (max_degree) *max_degree = w[1];
    return (std::size_t)w[0];
  }
  int32_t* block_sums = context.scratch<int32_t>(0, (std::size_t)grx::scan_num_blocks((int64_t)n) + 2);
  // (callers that do not ask for the longest row -- the merge-path advance itself -- keep the shorter sequence: no reset of the
  // maximum, no publishing launch; measured with them: merge_path 2.46 -> 2.69 ms for a BFS of the LJ stand-in)
  int* d_max = nullptr;
  if (max_degree) {
    d_max = context.scratch<int>(4, 4);
    error::throw_if_exception(hipMemsetAsync(d_max, 0, sizeof(int), s), "degrees");
  }
  std::size_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL((detail::degrees_kernel<graph_t, type_t, edge_t>), dim3((unsigned)g), dim3(256), 0, s, G, input,
                     n, seg, d_max);
  grx::exclusive_scan_i32(s, reinterpret_cast<const int32_t*>(seg), (int64_t)n, reinterpret_cast<int32_t*>(seg),
                          block_sums);
  if (host_words && max_degree) {
    hipLaunchKernelGGL((detail::publish_total_kernel<edge_t>), dim3(1), dim3(1), 0, s, seg + n, d_max, host_words);
    const int* w = context.wait_mailbox(4);
    if (max_degree) *max_degree = w[1];
    return (std::size_t)w[0];
  }
  if (max_degree) *max_degree = context.read_back(d_max)[0];
  return (std::size_t)context.read_back(reinterpret_cast<const int*>(seg + n))[0];
}

template <typename graph_t, typename frontier_t, typename work_tiles_t>
std::size_t compute_output_offsets(graph_t& G, frontier_t* input, work_tiles_t& segments,
                                   gcuda::standard_context_t& context, bool graph_as_frontier = false, int* max_degree = nullptr) {
  using type_t = typename frontier_t::type_t;
  if (graph_as_frontier)
    return compute_output_offsets<graph_t, type_t>(G, (const type_t*)nullptr,
                                                   (std::size_t)G.get_number_of_vertices(), segments, context, max_degree);
  return compute_output_offsets<graph_t, type_t>(G, input->data(), input->get_number_of_elements(), segments,
                                                 context, max_degree);
}

template <typename graph_t, typename frontier_t>
std::size_t compute_output_length(graph_t& G, frontier_t* input, gcuda::standard_context_t& context,
                                  bool graph_as_frontier = false) {
  using edge_t = typename graph_t::edge_type;
  if (graph_as_frontier) return (std::size_t)G.get_number_of_edges();
  const std::size_t n = input->get_number_of_elements();
  edge_t* tmp = context.scratch<edge_t>(1, n + 2);
  struct view_t {
    edge_t* p;
    std::size_t n;
    std::size_t size() const { return n; }
    void resize(std::size_t) {}
    edge_t* data() { return p; }
  } v{tmp, n + 2};
  return compute_output_offsets(G, input, v, context, false);
}

// reference spelling (helpers.hxx:127-131 takes the frontier by reference)
template <typename graph_t, typename frontier_t,
          typename = decltype(std::declval<frontier_t&>().get_number_of_elements())>
std::size_t compute_output_length(graph_t& G, frontier_t& input, gcuda::standard_context_t& context,
                                  bool graph_as_frontier = false) {
  return compute_output_length(G, &input, context, graph_as_frontier);
}

}  // namespace advance
}  // namespace operators
}  // namespace gunrock
