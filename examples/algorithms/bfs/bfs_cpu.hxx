// bfs_cpu.hxx -- host reference used by `bfs --validate`.
// Role parity: examples/algorithms/bfs/bfs_cpu.hxx:20-68 (reference), a
// priority-queue search with unit edge cost; same signature
// bfs_cpu::run<csr_t, vertex_t, edge_t>(csr, source, distances, predecessors) -> ms,
// same label for unreached vertices (numeric max), search-only timing.
// Written as a level-synchronous queue traversal (depths are identical; a queue
// is the natural host algorithm for unit costs).
#pragma once

#include <chrono>
#include <limits>
#include <vector>

#include <thrust/host_vector.h>

namespace bfs_cpu {

template <typename csr_t, typename vertex_t, typename edge_t>
float run(csr_t& csr, vertex_t& single_source, vertex_t* distances, vertex_t* predecessors) {
  thrust::host_vector<edge_t> offsets(csr.row_offsets);
  thrust::host_vector<vertex_t> targets(csr.column_indices);
  const vertex_t n = csr.number_of_rows;
  for (vertex_t v = 0; v < n; ++v) distances[v] = std::numeric_limits<vertex_t>::max();

  const auto t0 = std::chrono::high_resolution_clock::now();
  std::vector<vertex_t> queue;
  queue.reserve((std::size_t)n);
  distances[single_source] = 0;
  queue.push_back(single_source);
  for (std::size_t head = 0; head < queue.size(); ++head) {
    const vertex_t u = queue[head];
    const vertex_t next = distances[u] + 1;
    for (edge_t e = offsets[u]; e < offsets[u + 1]; ++e) {
      const vertex_t v = targets[e];
      if (next < distances[v]) {
        distances[v] = next;
        queue.push_back(v);
      }
    }
  }
  const auto t1 = std::chrono::high_resolution_clock::now();
  return (float)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count() / 1000;
}

}  // namespace bfs_cpu
