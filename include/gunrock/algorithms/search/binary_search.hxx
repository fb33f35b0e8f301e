// binary_search.hxx -- lower / upper bound over a sorted array, host and device.
// API parity: include/gunrock/algorithms/search/binary_search.hxx:31-60 (reference):
//   search::binary::execute(keys, key, begin, end, bound_t::{upper, lower}) -> first position in [begin, end) whose
//   element is > key (upper, the default) or >= key (lower).
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace search {

enum class bound_t { upper, lower };

namespace binary {

template <typename key_pointer_t, typename key_t, typename int_t>
__host__ __device__ __forceinline__ int_t execute(const key_pointer_t& keys, const key_t& key, int_t begin, int_t end,
                                                  const bound_t bounds = bound_t::upper) {
  int_t count = end - begin;  // invariant: the answer lies in [begin, begin + count]
  while (count > 0) {
    const int_t half = count / 2;
    const key_t probe = keys[begin + half];
    const bool go_right = (bounds == bound_t::upper) ? !(key < probe) : (probe < key);
    if (go_right) {
      begin += half + 1;
      count -= half + 1;
    } else {
      count = half;
    }
  }
  return begin;
}

}  // namespace binary
}  // namespace search
}  // namespace gunrock
