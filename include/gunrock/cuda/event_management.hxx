// event_management.hxx -- gcuda::event_t.
// Same include path as the reference (include/gunrock/cuda/event_management.hxx); the definitions live in <gunrock/cuda/context.hxx>.
#pragma once
#include <gunrock/cuda/context.hxx>
