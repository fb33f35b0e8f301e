// configs.hxx -- frontier kinds and views.
// API parity: include/gunrock/framework/frontier/configs.hxx:19-33 (reference).
#pragma once
namespace gunrock {
namespace frontier {
enum frontier_view_t { vector, bitmap, boolmap };
enum frontier_kind_t { vertex_frontier, edge_frontier, vertex_edge_frontier };
}  // namespace frontier
}  // namespace gunrock
