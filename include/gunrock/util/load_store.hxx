// load_store.hxx -- scalar global loads / stores used inside operators.
// API parity: include/gunrock/util/load_store.hxx:62-83 (reference), which wraps
// hipcub ThreadLoad/ThreadStore<DEFAULT>; on gfx950 a plain dereference emits the
// same global_load/global_store, so no vendor header is needed.
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace thread {

template <typename type_t>
__host__ __device__ __forceinline__ type_t load(const type_t* ptr) { return *ptr; }

template <typename type_t>
__host__ __device__ __forceinline__ void store(type_t* ptr, const type_t& value) { *ptr = value; }

}  // namespace thread
}  // namespace gunrock
