// formats.hxx -- umbrella.  API parity: include/gunrock/formats/formats.hxx (reference).
#pragma once
#include <gunrock/formats/coo.hxx>
#include <gunrock/formats/csr.hxx>
#include <gunrock/formats/csc.hxx>
