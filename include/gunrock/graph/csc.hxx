// csc.hxx -- the CSC view beside the CSR view (in-edge accessors, reverse_view()).
// Same include path as the reference (include/gunrock/graph/csc.hxx); the definitions live in <gunrock/graph/graph.hxx>.
#pragma once
#include <gunrock/graph/graph.hxx>
