// merge_path.hxx -- every workgroup gets the same number of edges (2048),
// whatever the degree distribution: hubs are split across workgroups, runs of
// tiny rows are fused.
// API parity: include/gunrock/framework/operators/advance/merge_path.hxx:78-362
// (reference): output = concatenation of neighbour lists in input order
// (output[global_atom]).  Differences: the reference gives each THREAD 11
// consecutive atoms (a wave then touches 64 addresses 44 bytes apart); here lanes
// take consecutive atoms so the column-index stream is coalesced.  The partition
// is on atoms only; slots are swept in LDS windows of 256, so runs of zero-degree
// slots cost extra windows instead of a second merge dimension.  Sizes are
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
  const edge_t a0 = (edge_t)blockIdx.x * detail::ATOMS_PER_BLOCK;
  const edge_t a1 = min(total_atoms, a0 + detail::ATOMS_PER_BLOCK);
  if (a0 >= a1) return;
  // first slot whose range contains atom a0: largest i with segments[i] <= a0.
  // 64 lanes probe 64 evenly spaced positions per round (k-ary search).
  if (threadIdx.x < 64) {
    std::size_t lo = 0, hi = n;  // invariant: segments[lo] <= a0 < segments[hi]
    while (hi - lo > 1) {
      const std::size_t span = hi - lo;
      const std::size_t step = (span + 63) / 64;
      const std::size_t pos = lo + (std::size_t)(threadIdx.x + 1) * step;
      const bool le = pos < hi && segments[pos] <= a0;
      const unsigned long long m = grx::dev::ballot(le);
      const int cnt = __popcll(m);  // positions are monotone => prefix of lanes
      const std::size_t nlo = lo + (std::size_t)cnt * step;
      const std::size_t nhi = (cnt < 64 && lo + (std::size_t)(cnt + 1) * step < hi) ? lo + (std::size_t)(cnt + 1) * step : hi;
      lo = nlo;
      hi = nhi;
    }
    if (threadIdx.x == 0) s_first = (int)lo;
  }
  __syncthreads();
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
  const std::size_t blocks = (total_atoms + detail::ATOMS_PER_BLOCK - 1) / detail::ATOMS_PER_BLOCK;
  hipLaunchKernelGGL((kernel<output_type, false, graph_t, operator_t, type_t, edge_t>), dim3((unsigned)blocks),
                     dim3(detail::BLOCK), 0, context.stream(), G, op, input, n, output, segments, (edge_t)total_atoms,
                     (int32_t*)nullptr);
}

// advance + compact filter in one pass: the kept neighbours land at output[0 .. *counter) (counter zeroed by the caller)
template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
void launch_compact(graph_t& G, operator_t op, const type_t* input, std::size_t n, type_t* output, const edge_t* segments,
                    std::size_t total_atoms, int32_t* counter, gcuda::standard_context_t& context) {
  if (n == 0 || total_atoms == 0) return;
  const std::size_t blocks = (total_atoms + detail::ATOMS_PER_BLOCK - 1) / detail::ATOMS_PER_BLOCK;
  hipLaunchKernelGGL((kernel<output_type, true, graph_t, operator_t, type_t, edge_t>), dim3((unsigned)blocks),
                     dim3(detail::BLOCK), 0, context.stream(), G, op, input, n, output, segments, (edge_t)total_atoms,
                     counter);
}

}  // namespace merge_path
}  // namespace advance
}  // namespace operators
}  // namespace gunrock
