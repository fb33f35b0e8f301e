// compare.hxx -- device-vs-host element comparison used by --validate.
// API parity: include/gunrock/util/compare.hxx:30-57 (reference): returns the
// number of mismatches, default comparator is exact `!=`.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <iostream>
#include <vector>

namespace gunrock {
namespace util {
namespace detail {
struct not_equal_t {
  template <typename a_t, typename b_t>
  bool operator()(const a_t& a, const b_t& b) const { return a != b; }
};
static const not_equal_t default_comparator{};
}  // namespace detail

template <typename type_t, typename comp_t = detail::not_equal_t>
std::size_t compare(const type_t* d_ptr, const type_t* h_ptr, const std::size_t n,
                    comp_t error_op = comp_t(), const bool verbose = false) {
  std::vector<type_t> from_device(n);
  if (n) (void)hipMemcpy(from_device.data(), d_ptr, n * sizeof(type_t), hipMemcpyDeviceToHost);
  std::size_t mismatches = 0;
  for (std::size_t i = 0; i < n; ++i) {
    if (error_op(from_device[i], h_ptr[i])) {
      if (verbose) std::cout << "Error: " << from_device[i] << " != " << h_ptr[i] << std::endl;
      ++mismatches;
    }
  }
  return mismatches;
}

}  // namespace util
}  // namespace gunrock
