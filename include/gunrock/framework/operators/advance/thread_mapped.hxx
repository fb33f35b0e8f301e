// thread_mapped.hxx -- one input vertex per thread, serial neighbour loop.
// API parity: include/gunrock/framework/operators/advance/thread_mapped.hxx:31-95
// (reference): neighbour k of input slot i lands at output[segments[i] + k].
// Best for uniformly low degrees (road networks); rows of 64+ neighbours are handed to the wave, of 2048+ to the workgroup.
#pragma once

#include <gunrock/framework/operators/advance/helpers.hxx>

namespace gunrock {
namespace operators {
namespace advance {
namespace thread_mapped {

// One input slot per thread, as upstream -- but a thread only walks rows of fewer than WAVE_ROW neighbours itself.  Longer
// rows are handed to the thread's WAVE (lanes on consecutive neighbours: coalesced column reads) and rows of BLOCK_ROW
// neighbours or more to the whole workgroup: upstream's kernel leaves a hub to one thread (thread_mapped.hxx:68-80), which
// on the LJ stand-in made the level behind the 125 k-edge source 90 ms of a 97 ms search.  Output positions are unchanged
// (neighbour k of slot i at segments[i] + k).
constexpr int WAVE_ROW = 64;
constexpr int BLOCK_ROW = 2048;

template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
__device__ __forceinline__ void visit(graph_t& G, operator_t& op, type_t v, edge_t e, edge_t out_at, type_t* output) {
  using vertex_t = typename graph_t::vertex_type;
  // mutable lvalues: user operators may take (vertex_t&, vertex_t&, edge_t const&, weight_t const&)
  // like the reference's hits.hxx:137
  vertex_t src = (vertex_t)v, nbr = G.get_destination_vertex(e);
  auto w = G.get_edge_weight(e);
  const bool keep = op(src, nbr, e, w);
  if constexpr (output_type != advance_io_type_t::none) {
    const type_t emitted = (output_type == advance_io_type_t::edges) ? (type_t)e : (type_t)nbr;
    output[out_at] = keep ? emitted : gunrock::numeric_limits<type_t>::invalid();
  }
}

template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
__global__ __launch_bounds__(256) void kernel(graph_t G, operator_t op, const type_t* input, std::size_t n,
                                              type_t* output, const edge_t* segments) {
  using vertex_t = typename graph_t::vertex_type;
  __shared__ int s_rows;               // rows handed to the workgroup
  __shared__ type_t s_v[256];
  __shared__ edge_t s_first[256], s_deg[256], s_base[256];
  const std::size_t i = (std::size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) s_rows = 0;
  __syncthreads();
  type_t v = gunrock::numeric_limits<type_t>::invalid();
  edge_t first = 0, deg = 0, base = 0;
  if (i < n) {
    v = input ? input[i] : (type_t)i;
    if (gunrock::util::limits::is_valid(v)) {
      first = G.get_starting_edge((vertex_t)v);
      deg = G.get_number_of_neighbors((vertex_t)v);
      base = segments[i];
    }
  }
  if (deg >= (edge_t)BLOCK_ROW) {
    const int at = atomicAdd(&s_rows, 1);
    s_v[at] = v; s_first[at] = first; s_deg[at] = deg; s_base[at] = base;
  } else if (deg < (edge_t)WAVE_ROW) {
    for (edge_t k = 0; k < deg; ++k) visit<output_type>(G, op, v, first + k, base + k, output);
  }
  // rows for the wave: every lane learns them from the owner's registers
  unsigned long long m = grx::dev::ballot(deg >= (edge_t)WAVE_ROW && deg < (edge_t)BLOCK_ROW);
  while (m) {
    const int l = __builtin_ctzll(m);
    m &= m - 1ull;
    const type_t rv = (type_t)__builtin_amdgcn_readlane((int)v, l);
    const edge_t rf = (edge_t)__builtin_amdgcn_readlane((int)first, l), rd = (edge_t)__builtin_amdgcn_readlane((int)deg, l),
                 rb = (edge_t)__builtin_amdgcn_readlane((int)base, l);
    for (edge_t k = (edge_t)lane; k < rd; k += 64) visit<output_type>(G, op, rv, rf + k, rb + k, output);
  }
  __syncthreads();
  const int rows = s_rows;
  for (int r = 0; r < rows; ++r) {
    const type_t rv = s_v[r];
    const edge_t rf = s_first[r], rd = s_deg[r], rb = s_base[r];
    for (edge_t k = (edge_t)threadIdx.x; k < rd; k += 256) visit<output_type>(G, op, rv, rf + k, rb + k, output);
  }
}

template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
void launch(graph_t& G, operator_t op, const type_t* input, std::size_t n, type_t* output, const edge_t* segments,
            gcuda::standard_context_t& context) {
  if (n == 0) return;
  hipLaunchKernelGGL((kernel<output_type, graph_t, operator_t, type_t, edge_t>), dim3((unsigned)((n + 255) / 256)),
                     dim3(256), 0, context.stream(), G, op, input, n, output, segments);
}

}  // namespace thread_mapped
}  // namespace advance
}  // namespace operators
}  // namespace gunrock
