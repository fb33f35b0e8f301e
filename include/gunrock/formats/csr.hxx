// csr.hxx -- compressed sparse row format, COO conversion, binary .csr files.
// API parity: include/gunrock/formats/csr.hxx:27-228 (reference): public fields,
// from_coo (stable row bucket sort: keeps duplicates, self loops and file order
// inside a row -- edge order is part of the contract),
// read_binary / write_binary ({rows:i32, cols:i32, nnz:i32} header then arrays).
#pragma once

#include <cstdio>
#include <string>
#include <type_traits>
#include <vector>

#include <grx.h>

#include <gunrock/container/vector.hxx>
#include <gunrock/error.hxx>
#include <gunrock/formats/coo.hxx>
#include <gunrock/memory.hxx>

namespace gunrock {
namespace format {

template <memory_space_t space, typename index_t, typename offset_t, typename value_t>
struct csr_t {
  using index_type = index_t;
  using offset_type = offset_t;
  using value_type = value_t;

  index_t number_of_rows = 0;
  index_t number_of_columns = 0;
  offset_t number_of_nonzeros = 0;
  vector_t<offset_t, space> row_offsets;
  vector_t<index_t, space> column_indices;
  vector_t<value_t, space> nonzero_values;

  csr_t() = default;
  csr_t(index_t r, index_t c, offset_t nnz)
      : number_of_rows(r), number_of_columns(c), number_of_nonzeros(nnz),
        row_offsets(r + 1), column_indices(nnz), nonzero_values(nnz) {}
  template <memory_space_t other>
  csr_t(const csr_t<other, index_t, offset_t, value_t>& rhs)
      : number_of_rows(rhs.number_of_rows), number_of_columns(rhs.number_of_columns),
        number_of_nonzeros(rhs.number_of_nonzeros), row_offsets(rhs.row_offsets),
        column_indices(rhs.column_indices), nonzero_values(rhs.nonzero_values) {}

  csr_t<space, index_t, offset_t, value_t> from_coo(
      const coo_t<memory_space_t::host, index_t, offset_t, value_t>& coo) {
    number_of_rows = coo.number_of_rows;
    number_of_columns = coo.number_of_columns;
    number_of_nonzeros = coo.number_of_nonzeros;
    const std::size_t R = (std::size_t)number_of_rows, NZ = (std::size_t)number_of_nonzeros;
    std::vector<offset_t> offsets(R + 1, 0);
    std::vector<index_t> cols(NZ);
    std::vector<value_t> vals(NZ);
    const index_t* I = coo.row_indices.data();
    const index_t* J = coo.column_indices.data();
    const value_t* X = coo.nonzero_values.data();
    for (std::size_t k = 0; k < NZ; ++k) ++offsets[(std::size_t)I[k] + 1];
    for (std::size_t r = 0; r < R; ++r) offsets[r + 1] += offsets[r];
    std::vector<offset_t> cursor(offsets.begin(), offsets.end() - 1);
    for (std::size_t k = 0; k < NZ; ++k) {
      const offset_t at = cursor[(std::size_t)I[k]]++;
      cols[(std::size_t)at] = J[k];
      vals[(std::size_t)at] = X[k];
    }
    row_offsets = thrust::host_vector<offset_t>(offsets.begin(), offsets.end());
    column_indices = thrust::host_vector<index_t>(cols.begin(), cols.end());
    nonzero_values = thrust::host_vector<value_t>(vals.begin(), vals.end());
    return *this;
  }

  // from_coo on the DEVICE (an extension: upstream converts on the host, formats/csr.hxx:81-140, and copies).  Device COO in,
  // this device CSR out, in the SAME order as the host builder -- stable by row: the entries of a row keep their input order,
  // duplicates and self loops included -- through the library's stable radix sort (grx_csr_from_coo_device, include/grx.h).
  template <memory_space_t s = space, typename std::enable_if<s == memory_space_t::device, int>::type = 0>
  csr_t<space, index_t, offset_t, value_t> from_coo(
      const coo_t<memory_space_t::device, index_t, offset_t, value_t>& coo) {
    static_assert(sizeof(index_t) == 4 && sizeof(offset_t) == 4 && std::is_same<value_t, float>::value,
                  "the device conversion handles 32-bit indices / offsets and float values");
    number_of_rows = coo.number_of_rows;
    number_of_columns = coo.number_of_columns;
    number_of_nonzeros = coo.number_of_nonzeros;
    row_offsets.resize((std::size_t)number_of_rows + 1);
    column_indices.resize((std::size_t)number_of_nonzeros);
    nonzero_values.resize((std::size_t)number_of_nonzeros);
    int device = 0;
    error::throw_if_exception(hipGetDevice(&device), "from_coo: hipGetDevice");
    error::throw_if_exception(hipDeviceSynchronize(), "from_coo: inputs pending");  // the triples may be in flight on any stream
    grx_context_t ctx = nullptr;
    error::throw_if_exception(grx_context_create(device, nullptr, &ctx) != GRX_SUCCESS, "from_coo: context");
    const grx_status_t st = grx_csr_from_coo_device(
        ctx, (int32_t)number_of_rows, (int64_t)number_of_nonzeros,
        reinterpret_cast<const int32_t*>(memory::raw_pointer_cast(coo.row_indices.data())),
        reinterpret_cast<const int32_t*>(memory::raw_pointer_cast(coo.column_indices.data())),
        memory::raw_pointer_cast(coo.nonzero_values.data()),
        reinterpret_cast<int32_t*>(memory::raw_pointer_cast(row_offsets.data())),
        reinterpret_cast<int32_t*>(memory::raw_pointer_cast(column_indices.data())),
        memory::raw_pointer_cast(nonzero_values.data()));
    const std::string why = st == GRX_SUCCESS ? std::string() : std::string(grx_last_error_string());
    (void)grx_context_destroy(ctx);
    error::throw_if_exception(st != GRX_SUCCESS, "from_coo (device): " + why);
    return *this;
  }

  void read_binary(std::string filename) {
    FILE* f = fopen(filename.c_str(), "rb");
    error::throw_if_exception(f == nullptr, "File could not be opened: " + filename);
    bool ok = fread(&number_of_rows, sizeof(index_t), 1, f) == 1 &&
              fread(&number_of_columns, sizeof(index_t), 1, f) == 1 &&
              fread(&number_of_nonzeros, sizeof(offset_t), 1, f) == 1;
    thrust::host_vector<offset_t> ro(ok ? (std::size_t)number_of_rows + 1 : 0);
    thrust::host_vector<index_t> ci(ok ? (std::size_t)number_of_nonzeros : 0);
    thrust::host_vector<value_t> nz(ok ? (std::size_t)number_of_nonzeros : 0);
    ok = ok && fread(ro.data(), sizeof(offset_t), ro.size(), f) == ro.size() &&
         fread(ci.data(), sizeof(index_t), ci.size(), f) == ci.size() &&
         fread(nz.data(), sizeof(value_t), nz.size(), f) == nz.size();
    fclose(f);
    error::throw_if_exception(!ok, "truncated binary csr: " + filename);
    row_offsets = ro;
    column_indices = ci;
    nonzero_values = nz;
  }

  void write_binary(std::string filename) {
    FILE* f = fopen(filename.c_str(), "wb");
    error::throw_if_exception(f == nullptr, "File could not be opened: " + filename);
    thrust::host_vector<offset_t> ro(row_offsets);
    thrust::host_vector<index_t> ci(column_indices);
    thrust::host_vector<value_t> nz(nonzero_values);
    fwrite(&number_of_rows, sizeof(index_t), 1, f);
    fwrite(&number_of_columns, sizeof(index_t), 1, f);
    fwrite(&number_of_nonzeros, sizeof(offset_t), 1, f);
    fwrite(ro.data(), sizeof(offset_t), ro.size(), f);
    fwrite(ci.data(), sizeof(index_t), ci.size(), f);
    fwrite(nz.data(), sizeof(value_t), nz.size(), f);
    fclose(f);
  }
};

}  // namespace format
}  // namespace gunrock
