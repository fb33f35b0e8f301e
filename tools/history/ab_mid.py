"""A/B of the two many-levels-per-launch bodies (grx_mid.hpp) on the road stand-in, one process:
    python tools/ab_mid.py [runs]
BFS (forward), SSSP with the graph's unit weights, SSSP with weights U{1..1000} (near-far); GRX_MID_VERSION=1|2 and
GRX_MID=0 (one launch pair per level).  Results of every configuration are compared with the first one's."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = WORKLOADS["road"]
src = (4894 // 2) * 4894 + 4894 // 2
ctx = gr.multi_context_t(0)
import ctypes as C  # noqa: E402
from gunrock_amd import _capi  # noqa: E402


def spare():
    out = (C.c_int32 * 5)()
    _capi.check(_capi.lib().grx_debug_ctrl(ctx._h, out, 5))
    return np.array(list(out), dtype=np.int64)


DEBUG = os.environ.get("GRX_MID_DEBUG") == "1"
ALGOS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bfs", "ssspu", "sssp"]
for algo in ALGOS:
    if algo == "sssp":
        props, csr = gr.generate("road", wl["V"], 0, wl["a"], 0.0, 1.0, seed=42)
    else:
        props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
    G = gr.build_graph(props, csr, ctx)
    V = G.get_number_of_vertices()
    ref = None
    for label, env in (("mid v2", {"GRX_MID_VERSION": "2"}), ("mid v1", {"GRX_MID_VERSION": "1"}), ("mid off", {"GRX_MID": "0"})):
        for k in ("GRX_MID_VERSION", "GRX_MID"):
            os.environ.pop(k, None)
        os.environ.update(env)
        if label == "mid off" and runs < 2:
            continue
        o = gr.options_t(advance_load_balance=gr.merge_path, advance_direction=gr.forward)
        d = torch.empty(V, dtype=torch.int32 if algo == "bfs" else torch.float32, device="cuda")
        times = []
        for _ in range(runs):
            s0 = spare() if DEBUG else None
            times.append(gr.bfs(G, src, d, None, ctx, o) if algo == "bfs" else gr.sssp(G, src, d, None, ctx, o))
            if DEBUG and label == "mid v2":
                ds = spare() - s0
                if algo == "bfs":
                    ds[0] += s0[0]  # the BFS seed kernel zeroes spare[0]
                lv = max(1, int(ds[4]))
                print("   phases, us per level over %d levels in multi-level launches: staged %.2f | ci+claims+compaction %.2f | flush %.2f | exchange %.2f"
                      % (lv, ds[0] * 0.01 / lv, ds[1] * 0.01 / lv, ds[2] * 0.01 / lv, ds[3] * 0.01 / lv), flush=True)
        st = gr.run_stats(ctx)
        h = d.cpu().numpy()
        if ref is None:
            ref = h.copy()
        print("%-6s %-8s ms %s | levels %d | us/level %.2f | same %s | edges %d" % (
            algo, label, [round(t, 2) for t in times], st["search_depth"], min(times) * 1e3 / max(1, st["search_depth"]),
            bool(np.array_equal(h, ref)), st["edges_visited"]), flush=True)
    del G
