// build.hxx -- graph::build<space>(properties, csr) -> graph view (by value).
// API parity: include/gunrock/graph/build.hxx:29-36 (reference).
#pragma once

#include <gunrock/formats/formats.hxx>
#include <gunrock/graph/graph.hxx>

namespace gunrock {
namespace graph {

template <memory_space_t space, typename vertex_t, typename edge_t, typename weight_t>
auto build(graph_properties_t properties, format::csr_t<space, vertex_t, edge_t, weight_t>& csr) {
  using view_t = graph_csr_t<space, vertex_t, edge_t, weight_t>;
  graph_t<space, vertex_t, edge_t, weight_t, view_t> G(properties);
  static_cast<view_t&>(G).set(csr);
  return G;
}

}  // namespace graph
}  // namespace gunrock
