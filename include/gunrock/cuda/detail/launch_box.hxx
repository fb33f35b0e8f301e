// launch_box.hxx -- launch-parameter pack selection (one architecture here: gfx950).
// Same include path as the reference (include/gunrock/cuda/detail/launch_box.hxx); the definitions live in <gunrock/cuda/launch_box.hxx>.
#pragma once
#include <gunrock/cuda/launch_box.hxx>
