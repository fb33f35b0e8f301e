// unique.hxx -- uniquify::unique::execute.
// Same include path as the reference (include/gunrock/framework/operators/uniquify/unique.hxx); the definitions live in <gunrock/framework/operators/uniquify/uniquify.hxx>.
#pragma once
#include <gunrock/framework/operators/uniquify/uniquify.hxx>
