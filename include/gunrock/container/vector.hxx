// vector.hxx -- vector_t<T, space>: the owning container of the public API.
// API parity: include/gunrock/container/vector.hxx:26-36 (reference): an alias of
// thrust::device_vector / thrust::host_vector, so user code keeps `.data().get()`,
// iterators and cross-space assignment.  (Containers are API surface; no
// operator on the hot path goes through thrust.)
#pragma once

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <gunrock/memory.hxx>

namespace gunrock {

template <typename type_t, memory_space_t space>
using vector_t = std::conditional_t<space == memory_space_t::host, thrust::host_vector<type_t>,
                                    thrust::device_vector<type_t>>;

}  // namespace gunrock
