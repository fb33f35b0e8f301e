// sssp_cpu.hxx -- host reference used by `sssp --validate`.
// Role parity: examples/algorithms/sssp/sssp_cpu.hxx:22-72 (reference): Dijkstra
// with a binary heap in weight_t arithmetic; same signature and labels
// (numeric max = unreached); search-only timing.
#pragma once

#include <chrono>
#include <functional>
#include <limits>
#include <queue>
#include <utility>
#include <vector>

#include <thrust/host_vector.h>

namespace sssp_cpu {

template <typename csr_t, typename vertex_t, typename edge_t, typename weight_t>
float run(csr_t& csr, vertex_t& single_source, weight_t* distances, vertex_t* predecessors) {
  thrust::host_vector<edge_t> offsets(csr.row_offsets);
  thrust::host_vector<vertex_t> targets(csr.column_indices);
  thrust::host_vector<weight_t> weights(csr.nonzero_values);
  const vertex_t n = csr.number_of_rows;
  for (vertex_t v = 0; v < n; ++v) distances[v] = std::numeric_limits<weight_t>::max();

  const auto t0 = std::chrono::high_resolution_clock::now();
  using entry_t = std::pair<weight_t, vertex_t>;  // (tentative distance, vertex)
  std::priority_queue<entry_t, std::vector<entry_t>, std::greater<entry_t>> heap;
  distances[single_source] = 0;
  heap.emplace((weight_t)0, single_source);
  while (!heap.empty()) {
    const entry_t top = heap.top();
    heap.pop();
    const vertex_t u = top.second;
    if (top.first > distances[u]) continue;  // stale entry
    for (edge_t e = offsets[u]; e < offsets[u + 1]; ++e) {
      const vertex_t v = targets[e];
      const weight_t through = top.first + weights[e];
      if (through < distances[v]) {
        distances[v] = through;
        heap.emplace(through, v);
      }
    }
  }
  const auto t1 = std::chrono::high_resolution_clock::now();
  return (float)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count() / 1000;
}

}  // namespace sssp_cpu
