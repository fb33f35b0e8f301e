// csr.hxx -- the CSR view (graph_csr_t accessors: get_starting_edge, get_number_of_neighbors, get_destination_vertex, get_edge_weight, get_source_vertex).
// Same include path as the reference (include/gunrock/graph/csr.hxx); the definitions live in <gunrock/graph/graph.hxx>.
#pragma once
#include <gunrock/graph/graph.hxx>
