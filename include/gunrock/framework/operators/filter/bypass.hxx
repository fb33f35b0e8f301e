// bypass.hxx -- filter::bypass::execute (out[i] = keep ? in[i] : -1, in place allowed).
// Same include path as the reference (include/gunrock/framework/operators/filter/bypass.hxx); the definitions live in <gunrock/framework/operators/filter/filter.hxx>.
#pragma once
#include <gunrock/framework/operators/filter/filter.hxx>
