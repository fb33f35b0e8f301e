// timer.hxx -- event-pair timer on a stream.
// API parity: include/gunrock/util/timer.hxx:18-61 (reference): timer_t with
// reset(), begin(stream), end(stream) -> elapsed ms (end synchronises).
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace util {

namespace detail {
// A kernel that does nothing, launched in front of a timer's start event (round 6).  An event recorded on an IDLE stream is
// stamped with the end of the stream's previous command, not with "now": whatever the host did in between -- the enactor's
// constructor allocating two frontier buffers, 44 ms of hipMalloc each on the LJ stand-in -- was then counted as GPU time of the
// next search (`GPU Elapsed Time : 38 ms` for a 4.6 ms block_mapped search, profiles/r6_c22_*, r6_c23_*).  With a command of
// its own in front of it the event is stamped when that command ends.
template <int = 0>
__global__ void timer_marker_kernel() {}
}  // namespace detail

struct timer_t {
  hipEvent_t start_ = nullptr, stop_ = nullptr;
  float time = 0.0f;

  timer_t() {
    (void)hipEventCreate(&start_);
    (void)hipEventCreate(&stop_);
  }
  ~timer_t() {
    if (start_) (void)hipEventDestroy(start_);
    if (stop_) (void)hipEventDestroy(stop_);
  }
  timer_t(const timer_t&) = delete;
  timer_t& operator=(const timer_t&) = delete;

  void reset() { time = 0.0f; }  // events are reusable; nothing to recreate
  void begin(hipStream_t stream = 0) {
    hipLaunchKernelGGL((detail::timer_marker_kernel<0>), dim3(1), dim3(1), 0, stream);
    (void)hipEventRecord(start_, stream);
  }
  float end(hipStream_t stream = 0) {
    (void)hipEventRecord(stop_, stream);
    (void)hipEventSynchronize(stop_);
    (void)hipEventElapsedTime(&time, start_, stop_);
    return milliseconds();
  }
  float seconds() const { return time * 1e-3f; }
  float milliseconds() const { return time; }
};

}  // namespace util
}  // namespace gunrock
