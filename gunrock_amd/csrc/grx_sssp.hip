// placeholder, replaced below
#include "grx_engine.hpp"
using namespace grx;
extern "C" grx_status_t grx_sssp(grx_context_t, grx_graph_t, int32_t, const grx_options_t*, float*, int32_t*, float*) {
  return fail(GRX_ERROR_UNSUPPORTED, "grx_sssp: not built yet");
}
