// frontier.hxx -- the vector frontier: a dense device array of vertex (or edge)
// ids in which -1 marks an empty slot, plus a host-tracked length.
// API parity: include/gunrock/framework/frontier/frontier.hxx:32-147 and
// vector_frontier.hxx:27-311 (reference): typedefs type_t/offset_t; host methods
// push_back, fill, sequence, resize, reserve, sort, print, data/begin/end,
// get_capacity, set_number_of_elements, set/get_resizing_factor, is_empty;
// host+device get_number_of_elements, get; device get_element_at/set_element_at.
// Copyable by value into kernels and lambdas (the copy shares the storage).
// Storage is a plain hipMalloc'ed buffer grown geometrically; growth preserves
// contents (reserve(size) allocates size * resizing_factor like the reference).
// The host mutators (resize, fill, sequence, push_back, growth) are HOST-SYNCHRONOUS in
// both directions like the reference's thrust calls: they wait for every stream of the
// device first (operators here run asynchronously on the context's non-blocking stream,
// which the null stream does not order against) and for their own kernel afterwards, so
// `f.resize(n); advance::execute(...)` and `advance::execute(...); f.fill(x)` are safe.
#pragma once

#include <hip/hip_runtime.h>
#include <thrust/device_ptr.h>
#include <thrust/sort.h>
#include <thrust/system/hip/execution_policy.h>

#include <iostream>
#include <memory>
#include <vector>

#include <gunrock/error.hxx>
#include <gunrock/framework/frontier/configs.hxx>
#include <gunrock/util/type_limits.hxx>

namespace gunrock {
namespace sort {
enum order_t { ascending, descending };
}
namespace frontier {
namespace detail {

template <typename type_t>
__global__ void fill_kernel(type_t* p, type_t value, std::size_t n) {
  for (std::size_t i = (std::size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * blockDim.x)
    p[i] = value;
}
template <typename type_t>
__global__ void sequence_kernel(type_t* p, type_t first, std::size_t n) {
  for (std::size_t i = (std::size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (std::size_t)gridDim.x * blockDim.x)
    p[i] = first + (type_t)i;
}
inline unsigned grid_for(std::size_t n) {
  std::size_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

template <typename type_t>
struct device_store_t {
  type_t* ptr = nullptr;
  std::size_t capacity = 0;
  ~device_store_t() {
    if (ptr) (void)hipFree(ptr);
  }
};

}  // namespace detail

template <typename vertex_t, typename edge_t, frontier_kind_t _kind = frontier_kind_t::vertex_frontier,
          frontier_view_t _view = frontier_view_t::vector>
class frontier_t {
 public:
  using vertex_type = vertex_t;
  using edge_type = edge_t;
  using type_t = std::conditional_t<_kind == frontier_kind_t::vertex_frontier, vertex_t, edge_t>;
  using offset_t = edge_t;
  using frontier_type = frontier_t<vertex_t, edge_t, _kind, _view>;
  static constexpr frontier_kind_t kind = _kind;
  static constexpr frontier_view_t view = _view;

  frontier_t() : store(std::make_shared<detail::device_store_t<type_t>>()) {}
  frontier_t(std::size_t size, float frontier_resizing_factor = 1.0f) : frontier_t() {
    resizing_factor = frontier_resizing_factor;
    resize(size);
  }

  // ---- host + device -------------------------------------------------------
  __host__ __device__ __forceinline__ std::size_t get_number_of_elements(hipStream_t = 0) const {
    return num_elements;
  }
  __host__ __device__ __forceinline__ type_t* get() const { return raw_ptr; }
  __host__ __device__ __forceinline__ constexpr frontier_kind_t get_kind() const { return _kind; }

  // ---- device --------------------------------------------------------------
  __device__ __forceinline__ type_t get_element_at(std::size_t const& idx) const noexcept { return raw_ptr[idx]; }
  __device__ __forceinline__ void set_element_at(type_t const& element, std::size_t const& idx) const noexcept {
    raw_ptr[idx] = element;
  }

  // ---- host ----------------------------------------------------------------
  std::size_t get_capacity() const { return store->capacity; }
  float get_resizing_factor() const { return resizing_factor; }
  void set_resizing_factor(float factor) { resizing_factor = factor; }
  void set_number_of_elements(std::size_t const& elements) { num_elements = elements; }
  type_t* data() { return raw_ptr; }
  type_t* begin() { return raw_ptr; }
  type_t* end() { return raw_ptr + num_elements; }
  bool is_empty() const { return num_elements == 0; }

  void reserve(std::size_t const& size) { grow((std::size_t)((double)size * resizing_factor)); }

  void resize(std::size_t const& size, type_t const default_value = gunrock::numeric_limits<type_t>::invalid()) {
    const std::size_t old = num_elements;
    grow(size);
    if (size > old) {
      quiesce();
      hipLaunchKernelGGL((detail::fill_kernel<type_t>), dim3(detail::grid_for(size - old)), dim3(256), 0, 0,
                         raw_ptr + old, default_value, size - old);
      error::throw_if_exception(hipStreamSynchronize(0), "frontier resize");
    }
    num_elements = size;
  }

  void push_back(type_t const& value) {
    if (num_elements + 1 > store->capacity) grow((num_elements + 1) * 2);
    quiesce();
    error::throw_if_exception(hipMemcpy(raw_ptr + num_elements, &value, sizeof(type_t), hipMemcpyHostToDevice),
                              "frontier push_back");
    ++num_elements;
  }

  void fill(type_t const value, hipStream_t stream = 0) {
    if (num_elements == 0) return;
    quiesce();
    hipLaunchKernelGGL((detail::fill_kernel<type_t>), dim3(detail::grid_for(num_elements)), dim3(256), 0, stream,
                       raw_ptr, value, num_elements);
    error::throw_if_exception(hipStreamSynchronize(stream), "frontier fill");
  }

  void sequence(type_t const initial_value, std::size_t const& size, hipStream_t stream = 0) {
    grow(size);
    num_elements = size;
    if (size == 0) return;
    quiesce();
    hipLaunchKernelGGL((detail::sequence_kernel<type_t>), dim3(detail::grid_for(size)), dim3(256), 0, stream,
                       raw_ptr, initial_value, size);
    error::throw_if_exception(hipStreamSynchronize(stream), "frontier sequence");
  }

  // Off the hot path (uniquify with full uniqueness only): vendor sort.
  void sort(sort::order_t order = sort::order_t::ascending, hipStream_t stream = 0) {
    if (num_elements < 2) return;
    thrust::device_ptr<type_t> b(raw_ptr);
    if (order == sort::order_t::ascending)
      thrust::sort(thrust::hip::par.on(stream), b, b + num_elements);
    else
      thrust::sort(thrust::hip::par.on(stream), b, b + num_elements, thrust::greater<type_t>());
  }

  void print() {
    std::vector<type_t> h(num_elements);
    if (num_elements)
      (void)hipMemcpy(h.data(), raw_ptr, num_elements * sizeof(type_t), hipMemcpyDeviceToHost);
    std::cout << "Frontier = ";
    for (auto& x : h) std::cout << x << " ";
    std::cout << std::endl;
  }

 private:
  // wait for pending operator work on every stream of the device (see the header comment)
  static void quiesce() { error::throw_if_exception(hipDeviceSynchronize(), "frontier sync"); }

  void grow(std::size_t want) {
    if (want <= store->capacity) return;
    quiesce();
    type_t* fresh = nullptr;
    error::throw_if_exception(hipMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(type_t)), "frontier alloc");
    if (store->ptr) {
      if (num_elements)
        error::throw_if_exception(
            hipMemcpy(fresh, store->ptr, num_elements * sizeof(type_t), hipMemcpyDeviceToDevice), "frontier copy");
      error::throw_if_exception(hipFree(store->ptr), "frontier free");
    }
    store->ptr = fresh;
    store->capacity = want;
    raw_ptr = fresh;
  }

  std::shared_ptr<detail::device_store_t<type_t>> store;
  type_t* raw_ptr = nullptr;
  std::size_t num_elements = 0;
  float resizing_factor = 1.0f;
};

}  // namespace frontier
}  // namespace gunrock
