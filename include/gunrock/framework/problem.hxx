// problem.hxx -- base class of an algorithm's data slice.
// API parity: include/gunrock/framework/problem.hxx:28-58 (reference): by-value
// graph view, shared multi-context, pure virtual init()/reset(), non-copyable.
#pragma once

#include <memory>

#include <gunrock/cuda/context.hxx>
#include <gunrock/graph/graph.hxx>

namespace gunrock {

template <typename graph_t>
struct problem_t {
  using vertex_t = typename graph_t::vertex_type;
  using edge_t = typename graph_t::edge_type;
  using weight_t = typename graph_t::weight_type;

  graph_t graph_slice;
  std::shared_ptr<gcuda::multi_context_t> context;

  problem_t() {}
  problem_t(graph_t& G, std::shared_ptr<gcuda::multi_context_t> _context) : graph_slice(G), context(_context) {}
  virtual ~problem_t() {}

  auto get_graph() { return graph_slice; }
  auto get_multi_context() { return context; }
  auto get_single_context(gcuda::device_id_t device = 0) { return context->get_context(device); }

  virtual void init() = 0;
  virtual void reset() = 0;

  problem_t(const problem_t&) = delete;
  problem_t& operator=(const problem_t&) = delete;
};

}  // namespace gunrock
