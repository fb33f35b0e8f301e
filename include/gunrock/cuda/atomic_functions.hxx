// atomic_functions.hxx -- atomicMin / atomicMax for every arithmetic type, floats included.
// API parity: include/gunrock/cuda/atomic_functions.hxx:22-124 (reference): gcuda::atomicMin / atomicMax(type_t*, type_t),
// returning the previous value.  The reference loops on atomicCAS for float and double; on gfx950 a float min / max is
// ONE native integer atomic on the ordered bit pattern (math::atomic::min / max, <gunrock/util/math.hxx>) and a double
// min / max is one 64-bit integer atomic chosen by the sign of the operand the same way.
#pragma once

#include <gunrock/util/math.hxx>

namespace gunrock {
namespace gcuda {

namespace detail {
// IEEE-754 order: for non-negative values the signed integer order of the bits is the value order; for negative values the
// unsigned order is the reverse value order.
__device__ __forceinline__ double atomic_min_f64(double* address, double value) {
  if (value >= 0.0)
    return __longlong_as_double(::atomicMin(reinterpret_cast<long long*>(address), __double_as_longlong(value)));
  return __longlong_as_double((long long)::atomicMax(reinterpret_cast<unsigned long long*>(address),
                                                      (unsigned long long)__double_as_longlong(value)));
}
__device__ __forceinline__ double atomic_max_f64(double* address, double value) {
  if (value >= 0.0)
    return __longlong_as_double(::atomicMax(reinterpret_cast<long long*>(address), __double_as_longlong(value)));
  return __longlong_as_double((long long)::atomicMin(reinterpret_cast<unsigned long long*>(address),
                                                      (unsigned long long)__double_as_longlong(value)));
}
}  // namespace detail

template <typename type_t>
__device__ __forceinline__ type_t atomicMin(type_t* address, type_t value) {
  if constexpr (std::is_same<type_t, double>::value) return detail::atomic_min_f64(address, value);
  else return math::atomic::min(address, value);
}

template <typename type_t>
__device__ __forceinline__ type_t atomicMax(type_t* address, type_t value) {
  if constexpr (std::is_same<type_t, double>::value) return detail::atomic_max_f64(address, value);
  else return math::atomic::max(address, value);
}

}  // namespace gcuda
}  // namespace gunrock
