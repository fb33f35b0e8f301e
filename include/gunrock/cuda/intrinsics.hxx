// intrinsics.hxx -- wave-level primitives (ballot, mbcnt rank, DPP scans) live in <gunrock/hip/wave.hxx>.
// Same include path as the reference (include/gunrock/cuda/intrinsics.hxx); the definitions live in <gunrock/hip/wave.hxx>.
#pragma once
#include <gunrock/hip/wave.hxx>
