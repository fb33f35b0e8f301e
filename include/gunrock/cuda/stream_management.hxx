// stream_management.hxx -- gcuda::stream_t.
// Same include path as the reference (include/gunrock/cuda/stream_management.hxx); the definitions live in <gunrock/cuda/context.hxx>.
#pragma once
#include <gunrock/cuda/context.hxx>
