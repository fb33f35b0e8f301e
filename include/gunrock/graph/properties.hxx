// properties.hxx -- graph_properties_t.
// API parity: include/gunrock/graph/properties.hxx:13-18 (reference), same defaults.
#pragma once
namespace gunrock {
namespace graph {
struct graph_properties_t {
  bool directed{false};
  bool weighted{true};
  bool symmetric{true};
  graph_properties_t() = default;
};
enum view_t : unsigned { csr = 1, csc = 2, coo = 4, invalid = 0 };
}  // namespace graph
}  // namespace gunrock
