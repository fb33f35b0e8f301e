"""Sub-phase clocks of the many-levels body (grx_mid.hpp) on the road stand-in, from the fine-timers build:
    python -m gunrock_amd.build --fine-timers
    GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_fine.so GRX_MID_DEBUG=1 python tools/mid_phases.py [unit|w|both]
Every sub-phase ends with a full wait of every thread, so what a phase issues is no longer hidden behind the next one: the
numbers are the latency of each dependent step as the leader workgroup sees it (microseconds per level, summed over the levels of
the multi-level launches), not a decomposition of the product build's level time (printed beside it from the default library's
4-phase clocks when GRX_LIB_PATH points at libgrx_timers.so)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from gunrock_amd import _capi  # noqa: E402
from bench import WORKLOADS  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
wl = WORKLOADS["road"]
side = int(os.environ.get("ROAD_AB_SIDE", "0")) or int(round(wl["V"] ** 0.5))
V = side * side
src = (side // 2) * side + side // 2
ctx = gr.multi_context_t(0)
NAMES = ["queue entry + row + label loaded", "block scan + LDS staged", "owner search (LDS)", "column indices (+ weights)",
         "claim + neighbours' row offsets", "second claim (stamp)", "codes, compaction, appends acked, side pile", "totals + exchange"]


def counters():
    out = (C.c_int32 * 13)()
    _capi.check(_capi.lib().grx_debug_ctrl(ctx._h, out, 13))
    return np.array(list(out), dtype=np.int64)


khz = 100000.0  # wall_clock64: 100 MHz
for weighted in ([False] if which == "unit" else [True] if which == "w" else [False, True]):
    props, csr = gr.generate("road", V, 0, wl["a"], wl["b"], 1.0 if weighted else wl["c"], seed=42)
    G = gr.build_graph(props, csr, ctx, device="cuda:0")
    d = torch.empty(V, dtype=torch.float32, device="cuda:0")
    o = gr.options_t(advance_load_balance=gr.merge_path)
    gr.sssp(G, src, d, None, ctx, o)
    c0 = counters()
    ms = gr.sssp(G, src, d, None, ctx, o)
    c1 = counters()
    st = gr.run_stats(ctx)
    dl = c1 - c0
    if not weighted:
        dl[0] = c1[0]  # (the BFS seed kernel zeroes spare[0])
    lv = max(1, int(dl[4]))
    print("== road %d x %d, %s: %.2f ms, %d levels (%d inside multi-level launches), %.2f us per level in this build" % (
        side, side, "U{1..1000} weights (near-far)" if weighted else "unit weights (BFS engine)", ms, st["search_depth"], lv,
        ms * 1e3 / max(1, st["search_depth"])))
    print("   4 phases, us per level: staged %.2f | chunk %.2f | totals %.2f | exchange %.2f" % tuple(dl[i] * 1e3 / khz / lv for i in range(4)))
    if dl[5:].sum() > 0:
        for i, n in enumerate(NAMES):
            print("   %-46s %6.2f us" % (n, dl[5 + i] * 1e3 / khz / lv))
        print("   %-46s %6.2f us" % ("sum", dl[5:].sum() * 1e3 / khz / lv))
    del G, d, csr
