// math.hxx -- atomics and small integer helpers usable from host and device.
// API parity: include/gunrock/util/math.hxx:75-136 + cuda/atomic_functions.hxx:22-44
// (reference): math::atomic::{add,min,max,cas,exch}, math::divide_round_up.
// gfx950 notes: float min/max are single integer atomics on the ordered bit
// pattern (the reference loops on CAS with fminf); float add is the native
// global_atomic_add_f32 (-munsafe-fp-atomics).  Host versions are plain
// read-modify-write, as in the reference.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace gunrock {
namespace math {

template <typename a_t, typename b_t>
__host__ __device__ __forceinline__ constexpr auto divide_round_up(a_t n, b_t d) {
  return (n + d - 1) / d;
}

namespace atomic {

template <typename type_t>
__host__ __device__ __forceinline__ type_t add(type_t* address, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(address, value);
#else
  type_t old = *address;
  *address = old + value;
  return old;
#endif
}

template <typename type_t>
__host__ __device__ __forceinline__ type_t min(type_t* address, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (std::is_same<type_t, float>::value) {
    if (value >= 0.0f)
      return __int_as_float(atomicMin(reinterpret_cast<int*>(address), __float_as_int(value)));
    return __uint_as_float(atomicMax(reinterpret_cast<unsigned*>(address), __float_as_uint(value)));
  } else {
    return atomicMin(address, value);
  }
#else
  type_t old = *address;
  *address = value < old ? value : old;
  return old;
#endif
}

template <typename type_t>
__host__ __device__ __forceinline__ type_t max(type_t* address, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (std::is_same<type_t, float>::value) {
    if (value >= 0.0f)
      return __int_as_float(atomicMax(reinterpret_cast<int*>(address), __float_as_int(value)));
    return __uint_as_float(atomicMin(reinterpret_cast<unsigned*>(address), __float_as_uint(value)));
  } else {
    return atomicMax(address, value);
  }
#else
  type_t old = *address;
  *address = value > old ? value : old;
  return old;
#endif
}

template <typename type_t>
__host__ __device__ __forceinline__ type_t cas(type_t* address, type_t compare, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (std::is_same<type_t, float>::value)
    return __int_as_float(atomicCAS(reinterpret_cast<int*>(address), __float_as_int(compare), __float_as_int(value)));
  else
    return atomicCAS(address, compare, value);
#else
  type_t old = *address;
  if (old == compare) *address = value;
  return old;
#endif
}

template <typename type_t>
__host__ __device__ __forceinline__ type_t exch(type_t* address, type_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicExch(address, value);
#else
  type_t old = *address;
  *address = value;
  return old;
#endif
}

}  // namespace atomic
}  // namespace math
}  // namespace gunrock
