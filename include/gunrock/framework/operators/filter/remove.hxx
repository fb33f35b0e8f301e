// remove.hxx -- filter::remove::execute (same stable compaction).
// Same include path as the reference (include/gunrock/framework/operators/filter/remove.hxx); the definitions live in <gunrock/framework/operators/filter/filter.hxx>.
#pragma once
#include <gunrock/framework/operators/filter/filter.hxx>
