// global.hxx -- thread / workgroup / grid index helpers.
// API parity: include/gunrock/cuda/global.hxx:15-103 (reference): gcuda::thread::{global,local}::id::{x,y,z},
// gcuda::block::{id,size}::{x,y,z[,total]}, gcuda::grid::size::{x,y,z,total}.  (On gfx950 a "block" is a workgroup of
// 64-lane waves; lane / wave helpers are in <gunrock/hip/wave.hxx>.)
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace gcuda {

typedef int thread_idx_t;

#define GUNROCK_XYZ(ns, X, Y, Z)                                   \
  namespace ns {                                                   \
  __device__ __forceinline__ int x() { return (int)(X); }          \
  __device__ __forceinline__ int y() { return (int)(Y); }          \
  __device__ __forceinline__ int z() { return (int)(Z); }          \
  }

namespace thread {
namespace global {  // index of the thread in the whole launch, per dimension
GUNROCK_XYZ(id, threadIdx.x + blockDim.x * blockIdx.x, threadIdx.y + blockDim.y * blockIdx.y,
            threadIdx.z + blockDim.z * blockIdx.z)
}  // namespace global
namespace local {   // index of the thread inside its workgroup
GUNROCK_XYZ(id, threadIdx.x, threadIdx.y, threadIdx.z)
}  // namespace local
}  // namespace thread

namespace block {
GUNROCK_XYZ(id, blockIdx.x, blockIdx.y, blockIdx.z)
namespace size {
__device__ __forceinline__ int x() { return (int)blockDim.x; }
__device__ __forceinline__ int y() { return (int)blockDim.y; }
__device__ __forceinline__ int z() { return (int)blockDim.z; }
__device__ __forceinline__ int total() { return x() * y() * z(); }
}  // namespace size
}  // namespace block

namespace grid {
namespace size {
__device__ __forceinline__ int x() { return (int)gridDim.x; }
__device__ __forceinline__ int y() { return (int)gridDim.y; }
__device__ __forceinline__ int z() { return (int)gridDim.z; }
__device__ __forceinline__ int total() { return x() * y() * z(); }
}  // namespace size
}  // namespace grid

#undef GUNROCK_XYZ

}  // namespace gcuda
}  // namespace gunrock
