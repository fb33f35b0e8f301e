// enactor.hxx -- the bulk-synchronous driver of an algorithm.
// API parity: include/gunrock/framework/enactor.hxx:31-344 (reference):
// enactor_properties_t{frontier_sizing_factor=1.5, number_of_frontier_buffers=2,
// self_manage_frontiers=false}; enactor_t<problem, kind, view> with public
// members properties/context/problem/frontiers/scanned_work_domain/
// active_frontier/inactive_frontier/buffer_selector/iteration, get_problem(),
// get_enactor(), get_input_frontier(), get_output_frontier(),
// swap_frontier_buffers(), float enact(); virtual loop (pure), prepare_frontier,
// is_converged (default: active frontier empty), finalize.
// Differences in behaviour (not in API): frontier buffers start at V elements
// and grow on demand instead of reserving 1.5 x max(E, V) up front, and enact()
// records the iteration count / runtime in benchmark::current().
#pragma once

#include <gunrock/util/trace.hxx>

#include <memory>
#include <vector>

#include <gunrock/container/vector.hxx>
#include <gunrock/cuda/cuda.hxx>
#include <gunrock/framework/benchmark.hxx>
#include <gunrock/framework/frontier/frontier.hxx>
#include <gunrock/framework/problem.hxx>

namespace gunrock {

struct enactor_properties_t {
  float frontier_sizing_factor{1.5f};
  std::size_t number_of_frontier_buffers{2};
  bool self_manage_frontiers{false};
  enactor_properties_t() = default;
};

template <typename algorithm_problem_t,
          frontier::frontier_kind_t frontier_kind = frontier::frontier_kind_t::vertex_frontier,
          frontier::frontier_view_t frontier_view = frontier::frontier_view_t::vector>
struct enactor_t {
  using vertex_t = typename algorithm_problem_t::vertex_t;
  using edge_t = typename algorithm_problem_t::edge_t;
  using frontier_t = frontier::frontier_t<vertex_t, edge_t, frontier_kind, frontier_view>;

  enactor_properties_t properties;
  std::shared_ptr<gcuda::multi_context_t> context;
  algorithm_problem_t* problem;
  std::vector<frontier_t> frontiers;
  vector_t<edge_t, memory_space_t::device> scanned_work_domain;
  frontier_t* active_frontier;
  frontier_t* inactive_frontier;
  int buffer_selector;
  int iteration;

  enactor_t(const enactor_t&) = delete;
  enactor_t& operator=(const enactor_t&) = delete;

  enactor_t(algorithm_problem_t* _problem, std::shared_ptr<gcuda::multi_context_t> _context,
            enactor_properties_t _properties = enactor_properties_t())
      : properties(_properties),
        context(_context),
        problem(_problem),
        frontiers(_properties.number_of_frontier_buffers < 2 ? 2 : _properties.number_of_frontier_buffers),
        scanned_work_domain((std::size_t)_problem->get_graph().get_number_of_vertices() + 1),
        active_frontier(&frontiers[0]),
        inactive_frontier(&frontiers[1]),
        buffer_selector(0),
        iteration(0) {
    if (!properties.self_manage_frontiers) {
      auto g = problem->get_graph();
      // as upstream (enactor.hxx:183-190): room for max(E, V) x frontier_sizing_factor elements per buffer, allocated HERE --
      // an advance without a filter writes one slot per out-edge of its input, and growing a buffer inside enact() is a device
      // synchronisation + hipMalloc + copy + hipFree per level (round 4: 8 of the 9.6 ms of `merge_path` without a filter on the
      // LJ stand-in were that)
      const std::size_t initial_size = (std::size_t)g.get_number_of_edges() > (std::size_t)g.get_number_of_vertices()
                                           ? (std::size_t)g.get_number_of_edges()
                                           : (std::size_t)g.get_number_of_vertices();
      for (auto& f : frontiers) {
        f.set_resizing_factor(properties.frontier_sizing_factor);
        f.reserve(initial_size);
      }
      // ... and the scanned degrees of an input frontier (one per slot + 1): an input holds at most the previous advance's output,
      // i.e. at most E slots.  Grown on demand (upstream: `scanned_work_domain(V)` and a resize inside compute_output_offsets) the
      // vector was re-allocated in the middle of every search whose level outgrew V slots -- hipMalloc + copy + hipFree of 145 MB,
      // 1.7-2.0 ms of the 3.1 ms of a `merge_path` BFS without a filter on the LJ stand-in (round 6, profiles/r6_c22_*).
      scanned_work_domain.resize(initial_size + 1);
    }
  }
  virtual ~enactor_t() {}

  algorithm_problem_t* get_problem() { return problem; }
  enactor_t* get_enactor() { return this; }
  frontier_t* get_input_frontier() { return active_frontier; }
  frontier_t* get_output_frontier() { return inactive_frontier; }

  void swap_frontier_buffers() {
    buffer_selector ^= 1;
    active_frontier = &frontiers[(std::size_t)buffer_selector];
    inactive_frontier = &frontiers[(std::size_t)(buffer_selector ^ 1)];
  }

  float enact() {
    iteration = 0;
    buffer_selector = 0;
    active_frontier = &frontiers[0];
    inactive_frontier = &frontiers[1];
    for (auto& f : frontiers) f.set_number_of_elements(0);

    auto single_context = context->get_context(0);
    auto& timer = single_context->timer();
    auto stream = single_context->stream();
    timer.reset();
    timer.begin(stream);
    GUNROCK_TRACE_RANGE("enact");
    prepare_frontier(get_input_frontier(), *context);
    while (!is_converged(*context)) {
#ifdef GUNROCK_ROCTX
      GUNROCK_TRACE_RANGE(std::string("iteration ") + std::to_string(iteration));
#endif
      loop(*context);
      ++iteration;
    }
    finalize(*context);
    const float runtime = timer.end(stream);
    benchmark::current().search_depth = (std::size_t)iteration;
    benchmark::current().total_runtime = runtime;
    return runtime;
  }

  virtual void loop(gcuda::multi_context_t& context) = 0;
  virtual void prepare_frontier(frontier_t* f, gcuda::multi_context_t& context) {}
  virtual bool is_converged(gcuda::multi_context_t& context) { return active_frontier->is_empty(); }
  virtual void finalize(gcuda::multi_context_t& context) {}
};

}  // namespace gunrock
