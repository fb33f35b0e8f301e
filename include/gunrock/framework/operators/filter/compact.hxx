// compact.hxx -- filter::compact::execute (single-pass wave-ballot compaction; throws upstream, real here).
// Same include path as the reference (include/gunrock/framework/operators/filter/compact.hxx); the definitions live in <gunrock/framework/operators/filter/filter.hxx>.
#pragma once
#include <gunrock/framework/operators/filter/filter.hxx>
