// vertex_pair.hxx -- {source, destination} of an edge as one 8-byte value.
// API parity: include/gunrock/graph/vertex_pair.hxx:12-23 (reference): graph::vertex_pair_t<vertex_t>,
// graph::edge_pair_t<edge_t>.
#pragma once

namespace gunrock {
namespace graph {

template <typename vertex_t>
struct alignas(8) vertex_pair_t {
  vertex_t source;
  vertex_t destination;
};

template <typename edge_t>
struct edge_pair_t {
  edge_t x;
  edge_t y;
};

}  // namespace graph
}  // namespace gunrock
