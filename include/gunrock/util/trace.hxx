// trace.hxx -- named ranges around operators and enactor iterations for rocprofv3's marker trace.
// The reference has no tracing hooks of its own ("tracing": nvtx/roctx absent; profiling is
// done from outside with the vendor profiler); the ranges make a `rocprofv3 --marker-trace --kernel-trace` timeline
// of a user's own algorithm readable: enact > iteration k > advance / filter / uniquify / parallel_for.
// Off by default (zero cost: the macro expands to nothing); build with -DGUNROCK_ROCTX and link -lroctx64.
#pragma once

#ifdef GUNROCK_ROCTX
#include <roctracer/roctx.h>

#include <string>

namespace gunrock {
namespace util {
struct trace_range_t {
  explicit trace_range_t(const char* name) { roctxRangePushA(name); }
  explicit trace_range_t(const std::string& name) { roctxRangePushA(name.c_str()); }
  trace_range_t(const trace_range_t&) = delete;
  ~trace_range_t() { roctxRangePop(); }
};
}  // namespace util
}  // namespace gunrock
#define GUNROCK_TRACE_CAT2(a, b) a##b
#define GUNROCK_TRACE_CAT(a, b) GUNROCK_TRACE_CAT2(a, b)
#define GUNROCK_TRACE_RANGE(name) ::gunrock::util::trace_range_t GUNROCK_TRACE_CAT(_gunrock_trace_, __LINE__)(name)
#else
#define GUNROCK_TRACE_RANGE(name) \
  do {                            \
  } while (0)
#endif
