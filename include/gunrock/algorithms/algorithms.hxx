// algorithms.hxx -- options shared by every algorithm + the umbrella include.
// API parity: include/gunrock/algorithms/algorithms.hxx:27-100 (reference):
// options_t with the same seven fields, defaults and constructors.  The extra
// `engine_flags` selects between the pre-compiled fused engine (libgrx.so, the
// default when linked) and the generic operator pipeline of these headers.
#pragma once

#include <gunrock/framework/operators/configs.hxx>

namespace gunrock {

struct options_t {
  operators::load_balance_t advance_load_balance = operators::load_balance_t::block_mapped;
  operators::filter_algorithm_t filter_algorithm = operators::filter_algorithm_t::predicated;
  bool enable_filter = false;
  bool enable_uniquify = false;
  operators::uniquify_algorithm_t uniquify_algorithm = operators::uniquify_algorithm_t::unique;
  bool best_effort_uniquify = true;
  float uniquify_percent = 100.0f;
  /// engine extension: 0 = fused engine when available; bit 0 = run the generic
  /// operator pipeline (advance -> filter -> uniquify as separate launches).
  int engine_flags = 0;
  /// engine extension: the reference declares advance_direction_t (operators/configs.hxx:78-82)
  /// but never reads it; `optimized` = direction-optimising BFS (bottom-up fat levels).
  operators::advance_direction_t advance_direction = operators::advance_direction_t::forward;

  options_t() = default;
  options_t(operators::load_balance_t _advance_load_balance,
            operators::filter_algorithm_t _filter_algorithm = operators::filter_algorithm_t::predicated,
            bool _enable_filter = false, bool _enable_uniquify = false,
            operators::uniquify_algorithm_t _uniquify_algorithm = operators::uniquify_algorithm_t::unique,
            bool _best_effort_uniquify = true, float _uniquify_percent = 100.0f)
      : advance_load_balance(_advance_load_balance), filter_algorithm(_filter_algorithm),
        enable_filter(_enable_filter), enable_uniquify(_enable_uniquify),
        uniquify_algorithm(_uniquify_algorithm), best_effort_uniquify(_best_effort_uniquify),
        uniquify_percent(_uniquify_percent) {}
};

}  // namespace gunrock

#include <gunrock/container/vector.hxx>
#include <gunrock/error.hxx>
#include <gunrock/formats/formats.hxx>
#include <gunrock/framework/framework.hxx>
#include <gunrock/graph/build.hxx>
#include <gunrock/graph/graph.hxx>
#include <gunrock/io/matrix_market.hxx>
#include <gunrock/memory.hxx>
#include <gunrock/util/compare.hxx>
#include <gunrock/util/math.hxx>
#include <gunrock/util/print.hxx>
#include <gunrock/algorithms/engine.hxx>
