// memory.hxx -- memory spaces and raw device allocation helpers.
// API parity: include/gunrock/memory.hxx (reference) -- memory_space_t{device,host},
// memory::allocate/free/copy, raw_pointer_cast.
#pragma once

#include <hip/hip_runtime.h>
#include <thrust/device_ptr.h>

#include <cstddef>
#include <gunrock/error.hxx>

namespace gunrock {
namespace memory {

enum memory_space_t { device, host };

template <typename type_t>
inline type_t* allocate(std::size_t bytes, memory_space_t space) {
  void* p = nullptr;
  if (bytes == 0) return nullptr;
  if (space == device)
    error::throw_if_exception(hipMalloc(&p, bytes), "hipMalloc failed");
  else
    error::throw_if_exception(hipHostMalloc(&p, bytes, hipHostMallocDefault), "hipHostMalloc failed");
  return reinterpret_cast<type_t*>(p);
}

template <typename type_t>
inline void free(type_t* p, memory_space_t space) {
  if (!p) return;
  if (space == device)
    error::throw_if_exception(hipFree(p), "hipFree failed");
  else
    error::throw_if_exception(hipHostFree(p), "hipHostFree failed");
}

template <typename type_t>
inline void copy(type_t* dst, const type_t* src, std::size_t count) {
  error::throw_if_exception(hipMemcpy(dst, src, count * sizeof(type_t), hipMemcpyDefault), "hipMemcpy failed");
}

template <typename type_t>
inline type_t* raw_pointer_cast(thrust::device_ptr<type_t> p) { return thrust::raw_pointer_cast(p); }
template <typename type_t>
__host__ __device__ inline type_t* raw_pointer_cast(type_t* p) { return p; }

}  // namespace memory
using memory::memory_space_t;
}  // namespace gunrock
