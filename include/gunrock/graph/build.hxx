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

// CSR + CSC views of the same graph (reference: graph/build.hxx:108-120): the CSR view stays the one every accessor and
// operator uses by default; the CSC view serves advance_direction_t::backward and the in-edge accessors.
template <memory_space_t space, typename vertex_t, typename edge_t, typename weight_t>
auto build(graph_properties_t properties, format::csr_t<space, vertex_t, edge_t, weight_t>& csr,
           format::csc_t<space, vertex_t, edge_t, weight_t>& csc) {
  using csr_view_t = graph_csr_t<space, vertex_t, edge_t, weight_t>;
  using csc_view_t = graph_csc_t<space, vertex_t, edge_t, weight_t>;
  graph_t<space, vertex_t, edge_t, weight_t, csr_view_t, csc_view_t> G(properties);
  static_cast<csr_view_t&>(G).set(csr);
  static_cast<csc_view_t&>(G).set(csc);
  return G;
}

}  // namespace graph
}  // namespace gunrock
