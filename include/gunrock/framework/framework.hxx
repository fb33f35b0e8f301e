// framework.hxx -- umbrella of the framework layer.
// API parity: include/gunrock/framework/framework.hxx (reference).
#pragma once
#include <gunrock/framework/benchmark.hxx>
#include <gunrock/framework/enactor.hxx>
#include <gunrock/framework/frontier/frontier.hxx>
#include <gunrock/framework/operators/operators.hxx>
#include <gunrock/framework/problem.hxx>
