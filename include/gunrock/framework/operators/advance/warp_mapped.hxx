// warp_mapped.hxx -- one 64-lane wavefront per input vertex; lanes stride the
// neighbour list, so column-index / weight reads of a row are fully coalesced.
// The reference only declares the enum (operators/configs.hxx:54) and throws
// "Load balance type not supported." (advance/advance.hxx:272-274); this is the
// real thing.  Best for medium-to-high uniform degrees.
#pragma once

#include <gunrock/framework/operators/advance/helpers.hxx>

namespace gunrock {
namespace operators {
namespace advance {
namespace warp_mapped {

template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
__global__ __launch_bounds__(256) void kernel(graph_t G, operator_t op, const type_t* input, std::size_t n,
                                              type_t* output, const edge_t* segments) {
  using vertex_t = typename graph_t::vertex_type;
  const int lane = grx::dev::lane_id();
  const std::size_t waves = ((std::size_t)gridDim.x * blockDim.x) >> 6;
  // A wave takes 64 consecutive slots at a time: the lanes load them (one coalesced read of the slots, their row offsets and
  // their output positions), a ballot keeps the slots that have neighbours, and the wave walks those one after the other with
  // the lanes on consecutive neighbours.  (Round 4: with one slot per wave iteration, the input of a BFS level WITHOUT a filter --
  // 36 M slots, 0.5 M of them valid, on the LJ stand-in -- cost one dependent load per slot and wave.)
  for (std::size_t i0 = (((std::size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64; i0 < n; i0 += waves * 64) {
    const std::size_t i = i0 + (std::size_t)lane;
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
    unsigned long long m = grx::dev::ballot(deg > 0);
    while (m) {
      const int l = __builtin_ctzll(m);
      m &= m - 1ull;
      const type_t rv = (type_t)__builtin_amdgcn_readlane((int)v, l);
      const edge_t rf = (edge_t)__builtin_amdgcn_readlane((int)first, l), rd = (edge_t)__builtin_amdgcn_readlane((int)deg, l),
                   rb = (edge_t)__builtin_amdgcn_readlane((int)base, l);
      for (edge_t k = lane; k < rd; k += 64) {
        // mutable lvalues: user operators may take (vertex_t&, vertex_t&, edge_t const&, weight_t const&)
        // like the reference's hits.hxx:137
        edge_t e = rf + k;
        vertex_t src = (vertex_t)rv, nbr = G.get_destination_vertex(e);
        auto w = G.get_edge_weight(e);
        const bool keep = op(src, nbr, e, w);
        if constexpr (output_type != advance_io_type_t::none) {
          const type_t emitted = (output_type == advance_io_type_t::edges) ? (type_t)e : (type_t)nbr;
          output[rb + k] = keep ? emitted : gunrock::numeric_limits<type_t>::invalid();
        }
      }
    }
  }
}

template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
void launch(graph_t& G, operator_t op, const type_t* input, std::size_t n, type_t* output, const edge_t* segments,
            gcuda::standard_context_t& context) {
  if (n == 0) return;
  std::size_t blocks = (n + 255) / 256;  // 4 waves per workgroup, 64 slots per wave and round
  const std::size_t cap = (std::size_t)context.props().multiProcessorCount * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL((kernel<output_type, graph_t, operator_t, type_t, edge_t>), dim3((unsigned)blocks), dim3(256), 0,
                     context.stream(), G, op, input, n, output, segments);
}

}  // namespace warp_mapped
}  // namespace advance
}  // namespace operators
}  // namespace gunrock
