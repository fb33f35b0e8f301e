// thread_mapped.hxx -- one input vertex per thread, serial neighbour loop.
// API parity: include/gunrock/framework/operators/advance/thread_mapped.hxx:31-95
// (reference): neighbour k of input slot i lands at output[segments[i] + k].
// Best for uniformly low degrees (road networks); a hub serialises its thread.
#pragma once

#include <gunrock/framework/operators/advance/helpers.hxx>

namespace gunrock {
namespace operators {
namespace advance {
namespace thread_mapped {

template <advance_io_type_t output_type, typename graph_t, typename operator_t, typename type_t, typename edge_t>
__global__ __launch_bounds__(256) void kernel(graph_t G, operator_t op, const type_t* input, std::size_t n,
                                              type_t* output, const edge_t* segments) {
  using vertex_t = typename graph_t::vertex_type;
  const std::size_t i = (std::size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const type_t v = input ? input[i] : (type_t)i;
  if (!gunrock::util::limits::is_valid(v)) return;
  const edge_t first = G.get_starting_edge((vertex_t)v);
  const edge_t deg = G.get_number_of_neighbors((vertex_t)v);
  const edge_t base = segments[i];
  for (edge_t k = 0; k < deg; ++k) {
    // mutable lvalues: user operators may take (vertex_t&, vertex_t&, edge_t const&, weight_t const&)
    // like the reference's hits.hxx:137
    edge_t e = first + k;
    vertex_t src = (vertex_t)v, nbr = G.get_destination_vertex(e);
    auto w = G.get_edge_weight(e);
    const bool keep = op(src, nbr, e, w);
    if constexpr (output_type != advance_io_type_t::none) {
      const type_t emitted = (output_type == advance_io_type_t::edges) ? (type_t)e : (type_t)nbr;
      output[base + k] = keep ? emitted : gunrock::numeric_limits<type_t>::invalid();
    }
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
