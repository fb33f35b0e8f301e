// cuda.hxx -- umbrella of the device-runtime layer ("gcuda").
// API parity: include/gunrock/cuda/cuda.hxx (reference).
#pragma once
#include <gunrock/cuda/context.hxx>
#include <gunrock/cuda/launch_box.hxx>
