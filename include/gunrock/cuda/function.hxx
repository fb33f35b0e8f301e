// function.hxx -- kernel attribute record.
// API parity: include/gunrock/cuda/function.hxx:17 (reference): gcuda::function_attributes_t.
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace gcuda {
typedef hipFuncAttributes function_attributes_t;
}  // namespace gcuda
}  // namespace gunrock
