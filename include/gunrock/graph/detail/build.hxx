// build.hxx -- graph::build / builder (graph/build.hxx here).
// Same include path as the reference (include/gunrock/graph/detail/build.hxx); the definitions live in <gunrock/graph/build.hxx>.
#pragma once
#include <gunrock/graph/build.hxx>
