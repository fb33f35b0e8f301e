// coo.hxx -- coordinate format.
// API parity: include/gunrock/formats/coo.hxx:23-46 (reference).
#pragma once

#include <gunrock/container/vector.hxx>

namespace gunrock {
namespace format {

template <memory_space_t space, typename index_t, typename nz_size_t, typename value_t>
struct coo_t {
  index_t number_of_rows = 0;
  index_t number_of_columns = 0;
  nz_size_t number_of_nonzeros = 0;
  vector_t<index_t, space> row_indices;
  vector_t<index_t, space> column_indices;
  vector_t<value_t, space> nonzero_values;

  coo_t() = default;
  coo_t(index_t r, index_t c, nz_size_t nnz)
      : number_of_rows(r), number_of_columns(c), number_of_nonzeros(nnz),
        row_indices(nnz), column_indices(nnz), nonzero_values(nnz) {}
  template <memory_space_t other>
  coo_t(const coo_t<other, index_t, nz_size_t, value_t>& rhs)
      : number_of_rows(rhs.number_of_rows), number_of_columns(rhs.number_of_columns),
        number_of_nonzeros(rhs.number_of_nonzeros), row_indices(rhs.row_indices),
        column_indices(rhs.column_indices), nonzero_values(rhs.nonzero_values) {}
};

}  // namespace format
}  // namespace gunrock
