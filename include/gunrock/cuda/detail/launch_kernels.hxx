// launch_kernels.hxx -- the generic blocked / strided index kernels.
// Same include path as the reference (include/gunrock/cuda/detail/launch_kernels.hxx); the definitions live in <gunrock/cuda/launch_box.hxx>.
#pragma once
#include <gunrock/cuda/launch_box.hxx>
