// launch_box.hxx -- launch-parameter carrier and the generic index kernels.
// API parity: include/gunrock/cuda/launch_box.hxx:112-360 and
// cuda/detail/launch_kernels.hxx:19-51 (reference): dim3_t<x,y,z>, launch_params_t /
// launch_params_dynamic_grid_t, launch_box_t::{launch_blocked, launch_strided,
// launch_cooperative, launch, calculate_grid_dimensions_*}, occupancy<launch_box_t>(kernel) and
// the `f(tid, bid, args...)`-style generic kernels.  The reference selects parameters per SM
// architecture at compile time; there is one architecture here (gfx950), so the first
// parameter set of the pack is the one used.
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

// static grid: launch_params_t<sm, block dims, grid dims, items per thread, shared bytes> (reference :88-110)
template <sm_flag_t sm_flags_, typename block_dimensions_, typename grid_dimensions_,
          std::size_t items_per_thread_ = 1, std::size_t shared_memory_bytes_ = 0>
struct launch_params_t : launch_params_dynamic_grid_t<sm_flags_, block_dimensions_, items_per_thread_, shared_memory_bytes_> {
  typedef grid_dimensions_ grid_dimensions_t;
  static constexpr bool static_grid = true;
};

namespace detail {
template <typename params_t, typename = void>
struct static_grid_of {
  static constexpr bool value = false;
  static dim3 get() { return dim3(1, 1, 1); }
};
template <typename params_t>
struct static_grid_of<params_t, std::void_t<typename params_t::grid_dimensions_t>> {
  static constexpr bool value = true;
  static dim3 get() { return params_t::grid_dimensions_t::get_dim3(); }
};
inline void collect_argument_addresses(void**) {}
template <typename first_t, typename... rest_t>
inline void collect_argument_addresses(void** out, first_t& first, rest_t&... rest) {
  *out = const_cast<void*>(static_cast<const void*>(&first));
  collect_argument_addresses(out + 1, rest...);
}
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
  dim3 block_dimensions = block_dimensions_t::get_dim3();
  dim3 grid_dimensions = detail::static_grid_of<params_t>::get();

  // reference :150-170 -- dynamic grids are sized from the element count; a static grid is kept
  void calculate_grid_dimensions_strided(std::size_t num_elements) {
    if (detail::static_grid_of<params_t>::value) return;
    const std::size_t g = math::divide_round_up(num_elements, (std::size_t)block_dimensions_t::size);
    grid_dimensions = dim3((unsigned)(g < 1 ? 1 : g), 1, 1);
  }
  void calculate_grid_dimensions_blocked(std::size_t num_elements) {
    if (detail::static_grid_of<params_t>::value) return;
    const std::size_t g = math::divide_round_up(num_elements, (std::size_t)block_dimensions_t::size * items_per_thread);
    grid_dimensions = dim3((unsigned)(g < 1 ? 1 : g), 1, 1);
  }

  // launch_cooperative(context, kernel, num_elements, args...): reference :293-307.  `kernel` is a
  // __global__ function; every workgroup of the grid must be co-resident for a grid-wide barrier to
  // complete, so a dynamic grid is clamped to what the device keeps resident.  On gfx950 / ROCm 7.2
  // the occupancy query can be one block per CU high at some SGPR counts and the cooperative launch
  // accepts the over-size grid (MI355X_MICROARCH.md, "Residency and cooperative launch"): one block
  // per CU is kept in reserve.
  template <typename func_t, typename... args_t>
  void launch_cooperative(standard_context_t& context, const func_t& f, const std::size_t num_elements,
                          args_t&&... args) {
    calculate_grid_dimensions_strided(num_elements);
    int per_cu = 0;
    error::throw_if_exception(hipOccupancyMaxActiveBlocksPerMultiprocessor(
                                  &per_cu, reinterpret_cast<const void*>(f), (int)block_dimensions_t::size,
                                  shared_memory_bytes),
                              "launch_cooperative: occupancy query");
    per_cu = per_cu > 8 ? 8 : per_cu;
    if (per_cu > 1) per_cu -= 1;
    const unsigned resident = (unsigned)(per_cu < 1 ? 1 : per_cu) * (unsigned)context.props().multiProcessorCount;
    if (!detail::static_grid_of<params_t>::value && grid_dimensions.x > resident) grid_dimensions.x = resident;
    constexpr std::size_t n_args = sizeof...(args_t) == 0 ? 1 : sizeof...(args_t);
    void* argument_ptrs[n_args] = {nullptr};
    detail::collect_argument_addresses(argument_ptrs, args...);
    error::throw_if_exception(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(f), grid_dimensions,
                                                         block_dimensions, argument_ptrs,
                                                         (unsigned)shared_memory_bytes, context.stream()),
                              "launch_cooperative");
  }

  // launch(context, kernel, args...): the kernel with this box's grid / block / shared memory (reference :327-335)
  template <typename func_t, typename... args_t>
  void launch(standard_context_t& context, const func_t& f, args_t&&... args) {
    hipLaunchKernelGGL(f, grid_dimensions, block_dimensions, (unsigned)shared_memory_bytes, context.stream(), args...);
  }

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

// Ratio of active to maximum waves per CU for `kernel` launched with this box's block size
// (reference :346-360).
template <typename launch_box_type, typename func_t>
inline float occupancy(func_t kernel) {
  int max_active_blocks = 0, device = 0;
  hipDeviceProp_t props;
  error::throw_if_exception(hipGetDevice(&device), "occupancy");
  error::throw_if_exception(hipGetDeviceProperties(&props, device), "occupancy");
  const int block_size = (int)launch_box_type::block_dimensions_t::size;
  error::throw_if_exception(
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_active_blocks, reinterpret_cast<const void*>(kernel), block_size,
                                                   (size_t)launch_box_type::shared_memory_bytes),
      "occupancy");
  return (float)(max_active_blocks * block_size / props.warpSize) /
         (float)(props.maxThreadsPerMultiProcessor / props.warpSize);
}

}  // namespace launch_box
}  // namespace gcuda
}  // namespace gunrock
