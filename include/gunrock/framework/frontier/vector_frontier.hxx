// vector_frontier.hxx -- the vector frontier (frontier_t with frontier_view_t::vector).
// Same include path as the reference (include/gunrock/framework/frontier/vector_frontier.hxx); the definitions live in <gunrock/framework/frontier/frontier.hxx>.
#pragma once
#include <gunrock/framework/frontier/frontier.hxx>
