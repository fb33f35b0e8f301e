// timer.hxx -- event-pair timer on a stream.
// API parity: include/gunrock/util/timer.hxx:18-61 (reference): timer_t with
// reset(), begin(stream), end(stream) -> elapsed ms (end synchronises).
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace util {

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
  void begin(hipStream_t stream = 0) { (void)hipEventRecord(start_, stream); }
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
