// base.hxx -- graph_base_t counts and properties (folded into graph_t here).
// Same include path as the reference (include/gunrock/graph/detail/base.hxx); the definitions live in <gunrock/graph/graph.hxx>.
#pragma once
#include <gunrock/graph/graph.hxx>
