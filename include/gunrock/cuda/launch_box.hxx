// launch_box.hxx -- launch-parameter carrier and the generic index kernels.
// API parity (shape only): include/gunrock/cuda/launch_box.hxx:112-335 and
// cuda/detail/launch_kernels.hxx:19-51 (reference): dim3_t<x,y,z>,
// launch_params_t, launch_box_t::{launch_blocked, launch_strided}, and the
// `f(tid, bid, args...)`-style generic kernels.  The reference selects
// parameters per SM architecture at compile time; there is one architecture
// here (gfx950), so the box only carries block size and items per thread.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <gunrock/cuda/context.hxx>
#include <gunrock/util/math.hxx>

namespace gunrock {
namespace gcuda {
namespace launch_box {

enum sm_flag_t : unsigned { fallback = ~0u, sm_gfx950 = 1u };

template <unsigned int x_, unsigned int y_ = 1, unsigned int z_ = 1>
struct dim3_t {
  enum : unsigned int { x = x_, y = y_, z = z_, size = x_ * y_ * z_ };
  static constexpr dim3 get_dim3() { return dim3(x_, y_, z_); }
};

template <sm_flag_t sm_flags_, typename block_dimensions_, std::size_t items_per_thread_ = 1,
          std::size_t shared_memory_bytes_ = 0>
struct launch_params_dynamic_grid_t {
  typedef block_dimensions_ block_dimensions_t;
  enum : unsigned { sm_flags = sm_flags_ };
  static constexpr std::size_t items_per_thread = items_per_thread_;
  static constexpr std::size_t shared_memory_bytes = shared_memory_bytes_;
};

namespace detail {
template <int threads, int items, typename func_t, typename... args_t>
__global__ __launch_bounds__(threads) void blocked_kernel(func_t f, const std::size_t bound, args_t... args) {
  const std::size_t base = ((std::size_t)blockIdx.x * threads + threadIdx.x) * items;
#pragma unroll
  for (int k = 0; k < items; ++k)
    if (base + k < bound) f((int)(base + k), (int)blockIdx.x, args...);
}
template <int threads, typename func_t, typename... args_t>
__global__ __launch_bounds__(threads) void strided_kernel(func_t f, const std::size_t bound, args_t... args) {
  for (std::size_t i = (std::size_t)blockIdx.x * threads + threadIdx.x; i < bound;
       i += (std::size_t)gridDim.x * threads)
    f((int)i, (int)blockIdx.x, args...);
}
}  // namespace detail

template <typename... lp_v>
struct launch_box_t {
  // the first (only meaningful) parameter set
  template <typename first_t, typename...>
  struct first_of { typedef first_t type; };
  typedef typename first_of<lp_v...>::type params_t;
  typedef typename params_t::block_dimensions_t block_dimensions_t;
  static constexpr std::size_t items_per_thread = params_t::items_per_thread;
  static constexpr std::size_t shared_memory_bytes = params_t::shared_memory_bytes;

  template <typename func_t, typename... args_t>
  void launch_blocked(standard_context_t& context, const func_t& f, const std::size_t num_elements,
                      args_t&&... args) {
    constexpr int T = block_dimensions_t::size;
    constexpr int I = (int)items_per_thread;
    const std::size_t grid = math::divide_round_up(num_elements, (std::size_t)T * I);
    if (grid == 0) return;
    hipLaunchKernelGGL((detail::blocked_kernel<T, I, func_t, std::decay_t<args_t>...>), dim3((unsigned)grid), dim3(T),
                       shared_memory_bytes, context.stream(), f, num_elements, args...);
  }

  template <typename func_t, typename... args_t>
  void launch_strided(standard_context_t& context, const func_t& f, const std::size_t num_elements,
                      args_t&&... args) {
    constexpr int T = block_dimensions_t::size;
    std::size_t grid = math::divide_round_up(num_elements, (std::size_t)T);
    const std::size_t cap = (std::size_t)context.props().multiProcessorCount * 8;
    if (grid > cap) grid = cap;
    if (grid == 0) return;
    hipLaunchKernelGGL((detail::strided_kernel<T, func_t, std::decay_t<args_t>...>), dim3((unsigned)grid), dim3(T),
                       shared_memory_bytes, context.stream(), f, num_elements, args...);
  }
};

}  // namespace launch_box
}  // namespace gcuda
}  // namespace gunrock
