// type_limits.hxx -- the "invalid" sentinel of frontier slots.
// API parity: include/gunrock/util/type_limits.hxx:17-71 (reference):
// numeric_limits<T>::invalid() is -1 for signed integers, max() for unsigned,
// NaN for floating point; util::limits::is_valid(x).
#pragma once

#include <cmath>
#include <limits>
#include <type_traits>

#include <hip/hip_runtime.h>

namespace gunrock {

template <typename type_t, typename enable = void>
struct numeric_limits : std::numeric_limits<type_t> {};

template <typename type_t>
struct numeric_limits<type_t, std::enable_if_t<std::is_integral<type_t>::value && std::is_signed<type_t>::value>>
    : std::numeric_limits<type_t> {
  __host__ __device__ static constexpr type_t invalid() { return static_cast<type_t>(-1); }
};

template <typename type_t>
struct numeric_limits<type_t, std::enable_if_t<std::is_integral<type_t>::value && std::is_unsigned<type_t>::value>>
    : std::numeric_limits<type_t> {
  __host__ __device__ static constexpr type_t invalid() { return std::numeric_limits<type_t>::max(); }
};

template <typename type_t>
struct numeric_limits<type_t, std::enable_if_t<std::is_floating_point<type_t>::value>>
    : std::numeric_limits<type_t> {
  __host__ __device__ static constexpr type_t invalid() { return std::numeric_limits<type_t>::quiet_NaN(); }
};

namespace util {
namespace limits {

template <typename type_t>
__host__ __device__ __forceinline__ bool is_valid(type_t value) {
  if constexpr (std::is_floating_point<type_t>::value)
    return !(value != value);  // NaN is the invalid marker
  else
    return value != gunrock::numeric_limits<type_t>::invalid();
}

}  // namespace limits
}  // namespace util
}  // namespace gunrock
