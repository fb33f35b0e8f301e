// csc.hxx -- compressed sparse column format: the in-edges of every vertex.
// API parity: include/gunrock/formats/csc.hxx:24-104 (reference): public fields
// number_of_rows / columns / nonzeros, column_offsets, row_indices, nonzero_values;
// from_csr (the reference sorts (column, row) pairs on the device with thrust; here a
// stable counting sort by column on the host -- entries of a column keep row order, which
// is the order the reference's sort produces).  SURVEY 8(f) f1: the view behind
// advance_direction_t::backward.
#pragma once

#include <vector>

#include <gunrock/container/vector.hxx>
#include <gunrock/formats/csr.hxx>

namespace gunrock {
namespace format {

template <memory_space_t space, typename index_t, typename offset_t, typename value_t>
struct csc_t {
  using index_type = index_t;
  using offset_type = offset_t;
  using value_type = value_t;

  index_t number_of_rows = 0;
  index_t number_of_columns = 0;
  offset_t number_of_nonzeros = 0;
  vector_t<offset_t, space> column_offsets;
  vector_t<index_t, space> row_indices;
  vector_t<value_t, space> nonzero_values;

  csc_t() = default;
  csc_t(index_t r, index_t c, offset_t nnz)
      : number_of_rows(r), number_of_columns(c), number_of_nonzeros(nnz),
        column_offsets(c + 1), row_indices(nnz), nonzero_values(nnz) {}
  template <memory_space_t other>
  csc_t(const csc_t<other, index_t, offset_t, value_t>& rhs)
      : number_of_rows(rhs.number_of_rows), number_of_columns(rhs.number_of_columns),
        number_of_nonzeros(rhs.number_of_nonzeros), column_offsets(rhs.column_offsets),
        row_indices(rhs.row_indices), nonzero_values(rhs.nonzero_values) {}

  // from a CSR in either memory space (a device CSR is copied to the host first)
  template <memory_space_t other>
  csc_t<space, index_t, offset_t, value_t> from_csr(const csr_t<other, index_t, offset_t, value_t>& csr) {
    number_of_rows = csr.number_of_rows;
    number_of_columns = csr.number_of_columns;
    number_of_nonzeros = csr.number_of_nonzeros;
    const std::size_t R = (std::size_t)number_of_rows, C = (std::size_t)number_of_columns, NZ = (std::size_t)number_of_nonzeros;
    thrust::host_vector<offset_t> ro = csr.row_offsets;
    thrust::host_vector<index_t> ci = csr.column_indices;
    thrust::host_vector<value_t> x = csr.nonzero_values;
    std::vector<offset_t> offsets(C + 1, 0);
    std::vector<index_t> rows(NZ);
    std::vector<value_t> vals(NZ);
    for (std::size_t k = 0; k < NZ; ++k) ++offsets[(std::size_t)ci[k] + 1];
    for (std::size_t c = 0; c < C; ++c) offsets[c + 1] += offsets[c];
    std::vector<offset_t> cursor(offsets.begin(), offsets.end() - 1);
    for (std::size_t r = 0; r < R; ++r)
      for (offset_t k = ro[r]; k < ro[r + 1]; ++k) {
        const offset_t at = cursor[(std::size_t)ci[(std::size_t)k]]++;
        rows[(std::size_t)at] = (index_t)r;
        vals[(std::size_t)at] = x.empty() ? value_t(1) : x[(std::size_t)k];
      }
    column_offsets = thrust::host_vector<offset_t>(offsets.begin(), offsets.end());
    row_indices = thrust::host_vector<index_t>(rows.begin(), rows.end());
    nonzero_values = thrust::host_vector<value_t>(vals.begin(), vals.end());
    return *this;
  }
};

}  // namespace format
}  // namespace gunrock
