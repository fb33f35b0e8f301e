// merge_path_v2.hxx -- load_balance_t::merge_path_v2 runs the merge_path kernel here (NVIDIA-only upstream).
// Same include path as the reference (include/gunrock/framework/operators/advance/merge_path_v2.hxx); the definitions live in <gunrock/framework/operators/advance/merge_path.hxx>.
#pragma once
#include <gunrock/framework/operators/advance/merge_path.hxx>
