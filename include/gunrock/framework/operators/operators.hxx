// operators.hxx -- umbrella.  API parity: include/gunrock/framework/operators/operators.hxx (reference).
#pragma once
#include <gunrock/framework/operators/advance/advance.hxx>
#include <gunrock/framework/operators/batch/batch.hxx>
#include <gunrock/framework/operators/configs.hxx>
#include <gunrock/framework/operators/filter/filter.hxx>
#include <gunrock/framework/operators/for/for.hxx>
#include <gunrock/framework/operators/uniquify/uniquify.hxx>
