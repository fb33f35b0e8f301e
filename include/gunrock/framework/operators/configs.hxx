// configs.hxx -- operator selection enums (unscoped; integer values are part of
// the contract: the reference's Python module exposes them as ints).
// API parity: include/gunrock/framework/operators/configs.hxx:52-112 (reference).
#pragma once

namespace gunrock {
namespace operators {

enum load_balance_t {
  thread_mapped,  // one input vertex per thread
  warp_mapped,    // one 64-lane wave per input vertex (enum-only in the reference)
  block_mapped,   // 256 input vertices per workgroup, edges shared through LDS
  bucketing,      // kernel chosen per frontier from its log2 degree histogram
  merge_path,     // equal number of edges per workgroup
  merge_path_v2,  // same kernel as merge_path here
  work_stealing   // not supported
};

enum advance_io_type_t { graph, vertices, edges, none };
enum advance_direction_t { forward, backward, optimized };
enum filter_algorithm_t { remove, predicated, compact, bypass };
enum uniquify_algorithm_t { unique, unique_copy };
enum parallel_for_each_t { vertex, edge, weight, element };

}  // namespace operators
}  // namespace gunrock
