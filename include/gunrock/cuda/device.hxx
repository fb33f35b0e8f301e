// device.hxx -- gcuda::device_id_t and device selection (standard_context_t owns the ordinal).
// Same include path as the reference (include/gunrock/cuda/device.hxx); the definitions live in <gunrock/cuda/context.hxx>.
#pragma once
#include <gunrock/cuda/context.hxx>
