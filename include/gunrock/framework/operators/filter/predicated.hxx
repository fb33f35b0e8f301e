// predicated.hxx -- filter::predicated::execute (stable compaction of valid && op).
// Same include path as the reference (include/gunrock/framework/operators/filter/predicated.hxx); the definitions live in <gunrock/framework/operators/filter/filter.hxx>.
#pragma once
#include <gunrock/framework/operators/filter/filter.hxx>
