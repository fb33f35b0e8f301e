// uniquify.hxx -- drop repeated elements of the active frontier.
// API parity: include/gunrock/framework/operators/uniquify/uniquify.hxx:26-94
// (reference): execute<type>(input*, output*, context, best_effort, percent) and
// execute<type = unique>(E, context, best_effort = false, percent = 100,
// swap_buffers = true).  Unless best_effort, the input is sorted first (100 %
// uniqueness); then one element of every run of equal neighbours is kept.
// The reference's unique variant swaps local pointer copies and then swaps the
// enactor buffers, leaving the NEXT operator on the stale buffer (unique.hxx:33-37,
// SURVEY 2.1); here the result is written to the output frontier, which becomes
// active after the swap.  Invalid (-1) slots are dropped as well.
#pragma once

#include <gunrock/util/trace.hxx>

#include <gunrock/cuda/context.hxx>
#include <gunrock/error.hxx>
#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/framework/frontier/frontier.hxx>
#include <gunrock/framework/operators/filter/filter.hxx>

namespace gunrock {
namespace operators {
namespace uniquify {
namespace detail {

// predicate over POSITIONS: keep slot i iff valid and different from slot i-1
template <typename type_t>
__global__ __launch_bounds__(256) void mark_runs_kernel(const type_t* in, std::size_t n, type_t* marked) {
  for (std::size_t i = (std::size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (std::size_t)gridDim.x * 256) {
    const type_t v = in[i];
    const bool first_of_run = (i == 0) || (in[i - 1] != v);
    marked[i] = (gunrock::util::limits::is_valid(v) && first_of_run) ? v : gunrock::numeric_limits<type_t>::invalid();
  }
}

struct keep_all_t {
  template <typename type_t>
  __host__ __device__ bool operator()(type_t const&) const { return true; }
};

}  // namespace detail

// Per-algorithm entry points with the reference's names (uniquify/unique.hxx:21-38, unique_copy.hxx:22-36):
// uniquify::<algorithm>::execute(input, output, standard_context) -- one element of every run of equal neighbours of
// `input` (invalid slots dropped), written to `output`.  Both names share one implementation; the headers of those names
// forward here.
namespace unique_copy {
template <typename frontier_t>
void execute(frontier_t* input, frontier_t* output, gcuda::standard_context_t& ctx) {
  using type_t = typename frontier_t::type_t;
  const std::size_t n = input->get_number_of_elements();
  if (output->get_capacity() < n) output->reserve(n);
  if (n == 0) {
    output->set_number_of_elements(0);
    return;
  }
  type_t* marked = ctx.template scratch<type_t>(3, n);
  hipLaunchKernelGGL((detail::mark_runs_kernel<type_t>), dim3(filter::detail::strided_grid(n, 256, ctx)), dim3(256), 0,
                     ctx.stream(), input->data(), n, marked);
  output->set_number_of_elements(
      filter::detail::stable_compact(detail::keep_all_t(), (const type_t*)marked, n, output->data(), ctx));
}
}  // namespace unique_copy

namespace unique {
template <typename frontier_t>
void execute(frontier_t* input, frontier_t* output, gcuda::standard_context_t& ctx) {
  unique_copy::execute(input, output, ctx);
}
}  // namespace unique

template <uniquify_algorithm_t type, typename frontier_t>
void execute(frontier_t* input, frontier_t* output, gcuda::multi_context_t& context,
             bool best_effort_uniquification = false, const float uniquification_percent = 100) {
  error::throw_if_exception(context.size() != 1, "`context.size() != 1` not supported");
  error::throw_if_exception(type != uniquify_algorithm_t::unique && type != uniquify_algorithm_t::unique_copy,
                            "Unique type not supported.");
  auto& ctx = *context.get_context(0);
  if (input->get_number_of_elements() != 0 && !best_effort_uniquification && uniquification_percent == 100)
    input->sort(sort::order_t::ascending, ctx.stream());
  if constexpr (type == uniquify_algorithm_t::unique) unique::execute(input, output, ctx);
  else unique_copy::execute(input, output, ctx);
}

template <uniquify_algorithm_t type = uniquify_algorithm_t::unique, typename enactor_type>
void execute(enactor_type* E, gcuda::multi_context_t& context, bool best_effort_uniquification = false,
             const float uniquification_percent = 100, bool swap_buffers = true) {
  if (!best_effort_uniquification)
    error::throw_if_exception(uniquification_percent < 0 || uniquification_percent > 100,
                              "Uniquification percentage must be a +ve float between 0 and 100.");
  execute<type>(E->get_input_frontier(), E->get_output_frontier(), context, best_effort_uniquification,
                uniquification_percent);
  if (swap_buffers) E->swap_frontier_buffers();
}

}  // namespace uniquify
}  // namespace operators
}  // namespace gunrock
