// unique_copy.hxx -- uniquify::unique_copy::execute.
// Same include path as the reference (include/gunrock/framework/operators/uniquify/unique_copy.hxx); the definitions live in <gunrock/framework/operators/uniquify/uniquify.hxx>.
#pragma once
#include <gunrock/framework/operators/uniquify/uniquify.hxx>
