// batch.hxx -- run an application several times with different inputs from host threads.
//
// Reference surface: operators::batch::execute(f, number_of_jobs, total_elapsed, args...)
// (include/gunrock/framework/operators/batch/batch.hxx:70-94 of the reference; used by
// ppr.hxx:235 and bc.hxx).  The reference starts one std::thread per job, all at once.
// Here a bounded pool of workers pulls job indices from an atomic counter: every job
// still runs on its own host thread context (its run() creates its own multi_context_t,
// i.e. its own HIP stream, so jobs overlap on the device), but 10^4 seeds do not mean
// 10^4 threads.  total_elapsed[0] receives the wall time of the whole batch in ms.
#pragma once

#include <gunrock/cuda/context.hxx>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstddef>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace gunrock {
namespace operators {
namespace batch {

template <typename function_t, typename... args_t>
void execute(function_t f, std::size_t number_of_jobs, float* total_elapsed, args_t&... /*args*/) {
  const auto t_start = std::chrono::steady_clock::now();
  std::vector<float> elapsed(number_of_jobs, 0.0f);
  std::atomic<std::size_t> next{0};
  std::exception_ptr failure;
  std::mutex failure_lock;
  const std::size_t hw = std::max<std::size_t>(1, std::thread::hardware_concurrency());
  const std::size_t n_workers = std::min<std::size_t>(number_of_jobs, std::min<std::size_t>(hw, 32));
  auto worker = [&]() {
    for (;;) {
      const std::size_t j = next.fetch_add(1);
      if (j >= number_of_jobs) return;
      try {
        elapsed[j] = f(j);
      } catch (...) {
        std::lock_guard<std::mutex> g(failure_lock);
        if (!failure) failure = std::current_exception();
      }
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(n_workers);
  for (std::size_t i = 0; i < n_workers; ++i) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
  const auto t_stop = std::chrono::steady_clock::now();
  if (total_elapsed)
    total_elapsed[0] = (float)std::chrono::duration_cast<std::chrono::microseconds>(t_stop - t_start).count() / 1000.0f;
  if (failure) std::rethrow_exception(failure);
}

}  // namespace batch
}  // namespace operators
}  // namespace gunrock
