// sample.hxx -- the 4 x 4 sample matrix of the reference's unit tests.
// API parity: include/gunrock/io/sample.hxx:47-91 (reference): io::sample::csr<space, vertex_t, edge_t, weight_t>()
//   rows 0..3: {}, {(0) 5, (1) 8}, {(2) 3}, {(1) 6}   ->  row_offsets 0 0 2 3 4 | columns 0 1 2 1 | values 5 8 3 6
// (SURVEY 8c golden vector (2); tests/cpp/test_host_utils.cu checks it.)
#pragma once

#include <gunrock/formats/formats.hxx>
#include <gunrock/graph/graph.hxx>

namespace gunrock {
namespace io {
namespace sample {

using namespace memory;

template <memory_space_t space = memory_space_t::device, typename vertex_t = int, typename edge_t = int,
          typename weight_t = float>
format::csr_t<space, vertex_t, edge_t, weight_t> csr() {
  format::csr_t<memory_space_t::host, vertex_t, edge_t, weight_t> h(4, 4, 4);
  const edge_t offsets[5] = {0, 0, 2, 3, 4};
  const vertex_t columns[4] = {0, 1, 2, 1};
  const weight_t values[4] = {5, 8, 3, 6};
  for (int i = 0; i < 5; ++i) h.row_offsets[i] = offsets[i];
  for (int i = 0; i < 4; ++i) {
    h.column_indices[i] = columns[i];
    h.nonzero_values[i] = values[i];
  }
  if constexpr (space == memory_space_t::host) {
    return h;
  } else {
    format::csr_t<memory_space_t::device, vertex_t, edge_t, weight_t> d(h);
    return d;
  }
}

}  // namespace sample
}  // namespace io
}  // namespace gunrock
