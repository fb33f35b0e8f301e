// benchmark.hxx -- per-run traversal counters of the metrics build.
// API parity: include/gunrock/framework/benchmark.hxx:43-98 (reference):
// host_benchmark_t, INIT_BENCH / EXTRACT / DESTROY_BENCH.  The reference counts
// with a device atomicAdd per frontier element inside its kernels (and that
// configuration does not compile upstream); here the counters
// are accumulated by the operators from the scan totals they already compute,
// so collection is always on and free.
#pragma once

#include <cstddef>

namespace gunrock {
namespace benchmark {

struct host_benchmark_t {
  unsigned int edges_visited = 0;
  unsigned int vertices_visited = 0;
  std::size_t search_depth = 0;
  double total_runtime = 0;
};

inline host_benchmark_t& current() {
  static host_benchmark_t instance;
  return instance;
}
inline void INIT_BENCH() { current() = host_benchmark_t(); }
inline host_benchmark_t EXTRACT() { return current(); }
inline void DESTROY_BENCH() {}
inline void LOG_EDGES_HOST(std::size_t n) { current().edges_visited += (unsigned int)n; }
inline void LOG_VERTICES_HOST(std::size_t n) { current().vertices_visited += (unsigned int)n; }

}  // namespace benchmark
}  // namespace gunrock
