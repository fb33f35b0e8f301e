// merge_path.hxx -- every workgroup gets the same number of input slots + edges
// (2048), whatever the degree distribution: hubs are split across workgroups,
// runs of tiny rows are fused, runs of invalid / empty slots are shared out.
// API parity: include/gunrock/framework/operators/advance/merge_path.hxx:78-362
// (reference): output = concatenation of neighbour lists in input order
// (output[global_atom]).  Differences: the reference gives each THREAD 11
// consecutive atoms (a wave then touches 64 addresses 44 bytes apart); here lanes
// take consecutive atoms so the column-index stream is coalesced.  The partition
// is on the merge path of slots and atoms (found per workgroup by a k-ary search
// over the scanned degrees); inside a workgroup slots are swept in LDS windows of 256.  Sizes are
// 64-bit on the host side; 32-bit atom ids inside a launch (E < 2^31).
#pragma once

#include <gunrock/framework/operators/advance/helpers.hxx>

namespace gunrock {
namespace operators {
namespace advance {
namespace merge_path {

template <advance_io_type_t output_type, bool COMPACT = false, typename graph_t, typename operator_t, typename type_t,
          typename edge_t>
__global__ __launch_bounds__(detail::BLOCK) void kernel(graph_t G, operator_t op, const type_t* input,
                                                        std::size_t n, type_t* output, const edge_t* segments,
                                                        edge_t total_atoms, int32_t* counter = nullptr) {
  using vertex_t = typename graph_t::vertex_type;
  __shared__ int s_cmp[detail::BLOCK / 64 + 2];
  __shared__ int s_seg[detail::BLOCK + 1];
  __shared__ int s_start[detail::BLOCK];
  __shared__ type_t s_src[detail::BLOCK];
  __shared__ int s_first;
  __shared__ long long s_a[2];
  // Partition on the MERGE PATH of slots and atoms (round 4): workgroup b owns the stretch [b, b + 1) x ITEMS of the diagonal
  // d = slots passed + atoms consumed; a state of the path is (i, d - i) with i the largest slot index for which
  // i + segments[i] <= d (the sum is strictly increasing in i).  Before, the partition was on atoms alone and a workgroup swept
  // however many slots its 2048 atoms were spread over: on the input of a BFS level WITHOUT a filter -- 36 M slots, 0.5 M of them
  // valid, 0.7 M atoms on the LJ stand-in -- that was 390 windows of 256 slots per workgroup, 3 ms for the level (upstream's
  // merge path splits both dimensions, merge_path.hxx:112-362).  Now a workgroup sweeps at most ITEMS slots.
  // Waves 0 and 1 find the two ends: 64 lanes probe 64 evenly spaced positions per round (k-ary search).
  const unsigned long long d_total = (unsigned long long)n + (unsigned long long)total_atoms;
  const unsigned long long d_lo = (unsigned long long)blockIdx.x * detail::ATOMS_PER_BLOCK;
  if (d_lo >= d_total) return;
  const unsigned long long d_hi = d_lo + detail::ATOMS_PER_BLOCK < d_total ? d_lo + detail::ATOMS_PER_BLOCK : d_total;
  if (threadIdx.x < 128) {
    const int lane = threadIdx.x & 63;
    const unsigned long long d = threadIdx.x < 64 ? d_lo : d_hi;
    std::size_t lo = 0, hi = n + 1;  // invariant: lo + segments[lo] <= d; hi == n + 1 or hi + segments[hi] > d
    while (hi - lo > 1) {
      const std::size_t span = hi - lo;
      const std::size_t step = (span + 63) / 64;
      const std::size_t pos = lo + (std::size_t)(lane + 1) * step;
      const bool le = pos < hi && (unsigned long long)pos + (unsigned long long)segments[pos] <= d;
      const unsigned long long m = grx::dev::ballot(le);
      const int cnt = __popcll(m);  // positions are monotone => prefix of lanes
      const std::size_t nlo = lo + (std::size_t)cnt * step;
      const std::size_t nhi = (cnt < 64 && lo + (std::size_t)(cnt + 1) * step < hi) ? lo + (std::size_t)(cnt + 1) * step : hi;
      lo = nlo;
      hi = nhi;
    }
    if (lane == 0) {
      if (threadIdx.x == 0) { s_first = (int)lo; s_a[0] = (long long)(d - lo); }
      else s_a[1] = (long long)(d - lo);
    }
  }
  __syncthreads();
  const edge_t a0 = (edge_t)min((long long)total_atoms, s_a[0]);
  const edge_t a1 = (edge_t)min((long long)total_atoms, s_a[1]);
  if (a0 >= a1) return;  // a stretch of empty slots
  for (std::size_t ws = (std::size_t)s_first; ws < n; ws += detail::BLOCK) {
    const std::size_t i = ws + threadIdx.x;
    const edge_t base = segments[ws];
    if (base >= a1) break;
    type_t v = gunrock::numeric_limits<type_t>::invalid();
    int start = 0;
    edge_t rel = 0;
    if (i <= n) rel = segments[i < n ? i : n] - base;
    if (i < n) {
      v = input ? input[i] : (type_t)i;
      if (gunrock::util::limits::is_valid(v)) start = (int)G.get_starting_edge((vertex_t)v);
    }
    const std::size_t last = (ws + detail::BLOCK < n) ? ws + detail::BLOCK : n;
    const int nslots = (int)(last - ws);
    __syncthreads();  // previous window fully consumed
    if ((int)threadIdx.x < nslots) {
      s_seg[threadIdx.x] = (int)rel;
      s_start[threadIdx.x] = start;
      s_src[threadIdx.x] = v;
    }
    if (threadIdx.x == 0) s_seg[nslots] = (int)(segments[last] - base);
    __syncthreads();
    const int window_total = s_seg[nslots];
    const int lo = (int)max((edge_t)0, a0 - base);
    const int hi = (int)min((edge_t)window_total, a1 - base);
    type_t* out = nullptr;
    if constexpr (COMPACT) out = output;  // positions come from the counter, not from the scan
    else if constexpr (output_type != advance_io_type_t::none) out = output + base;
    if (lo < hi) detail::expand_window<output_type, COMPACT>(G, op, s_seg, s_start, s_src, nslots, lo, hi, out, s_cmp, counter);
    if (base + window_total >= a1) break;
  }
}

template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
void launch(graph_t& G, operator_t op, const type_t* input, std::size_t n, type_t* output, const edge_t* segments,
            std::size_t total_atoms, gcuda::standard_context_t& context) {
  if (n == 0 || total_atoms == 0) return;
  const std::size_t blocks = (n + total_atoms + detail::ATOMS_PER_BLOCK - 1) / detail::ATOMS_PER_BLOCK;
  hipLaunchKernelGGL((kernel<output_type, false, graph_t, operator_t, type_t, edge_t>), dim3((unsigned)blocks),
                     dim3(detail::BLOCK), 0, context.stream(), G, op, input, n, output, segments, (edge_t)total_atoms,
                     (int32_t*)nullptr);
}

// advance + compact filter in one pass: the kept neighbours land at output[0 .. *counter) (counter zeroed by the caller)
template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
void launch_compact(graph_t& G, operator_t op, const type_t* input, std::size_t n, type_t* output, const edge_t* segments,
                    std::size_t total_atoms, int32_t* counter, gcuda::standard_context_t& context) {
  if (n == 0 || total_atoms == 0) return;
  const std::size_t blocks = (n + total_atoms + detail::ATOMS_PER_BLOCK - 1) / detail::ATOMS_PER_BLOCK;
  hipLaunchKernelGGL((kernel<output_type, true, graph_t, operator_t, type_t, edge_t>), dim3((unsigned)blocks),
                     dim3(detail::BLOCK), 0, context.stream(), G, op, input, n, output, segments, (edge_t)total_atoms,
                     counter);
}

}  // namespace merge_path
}  // namespace advance
}  // namespace operators
}  // namespace gunrock
