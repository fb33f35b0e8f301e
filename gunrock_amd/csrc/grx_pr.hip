#include "grx_engine.hpp"
using namespace grx;
extern "C" grx_status_t grx_pr(grx_context_t, grx_graph_t, float, float, const grx_options_t*, float*, int32_t*, float*) {
  return fail(GRX_ERROR_UNSUPPORTED, "grx_pr: not built yet");
}
