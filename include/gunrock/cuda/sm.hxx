// sm.hxx -- launch_box::sm_flag_t (one architecture here: sm_gfx950, and `fallback`).
// Same include path as the reference (include/gunrock/cuda/sm.hxx); the definitions live in <gunrock/cuda/launch_box.hxx>.
#pragma once
#include <gunrock/cuda/launch_box.hxx>
