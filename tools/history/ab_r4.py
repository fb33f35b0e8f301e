"""Round-4 A/B of the binned forward levels in ONE process per graph: sub-counters per bin in the scatter's histogram
(GRX_BIN_SUB), pair stores in its copy-out (GRX_BIN_PAIR).
    python tools/ab_r4.py [lj|kron|twitter] [reps]        -- same protocol and output as tools/ab_r3.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
V = G.get_number_of_vertices()
d = torch.empty(V, dtype=torch.int32, device="cuda")
KNOBS = ("GRX_BIN_PAIR", "GRX_BIN_E16", "GRX_SC2_STATIC", "GRX_BIN_SUB")
ref = None


def run(label, direction, env=None):
    global ref
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                     advance_direction=direction, engine_flags=gr.FLAG_ASYNC_RETURN)
    for _ in range(3):
        gr.bfs(G, src, d, None, ctx, o)
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gr.bfs(G, src, d, None, ctx, o)
    ctx.synchronize()
    step = (time.perf_counter() - t0) * 1e3 / reps
    st = gr.run_stats(ctx)
    h = d.cpu().numpy()
    if ref is None:
        ref = h.copy()
    same = bool(np.array_equal(h, ref))
    po = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                      advance_direction=direction, engine_flags=gr.FLAG_PROFILE)
    best = None
    for _ in range(3):
        gr.bfs(G, src, d, None, ctx, po)
        prof = gr.level_profile(ctx)
        t = sum(l["advance_ms"] for l in prof)
        if best is None or t < best[0]:
            best = (t, prof)
    same = same and bool(np.array_equal(d.cpu().numpy(), ref))
    fat = sorted(best[1], key=lambda l: -l["edges"])[:2]
    frac = sum(12 * l["frontier_size"] + 12 * l["edges"] for l in fat) / max(1e-9, sum(l["advance_ms"] for l in fat) * 1e-3) / 8e12
    lv = " ".join("%d/%d:%s%.0f+h%.0f" % (l["frontier_size"], l["edges"], {0: "T", 1: "B", 2: "N", 3: "M"}.get(l.get("bottom_up"), "?"),
                                          l["advance_ms"] * 1e3, l["other_ms"] * 1e3) for l in best[1])
    print("%-38s step %.4f ms | enact %.4f | GTEPS %.1f | fat-levels frac %.3f | same %s | %s"
          % (label, step, st["elapsed_ms"], st["edges_visited"] / (step * 1e6), frac, same, lv), flush=True)


print("workload", name, "V", V, "E", G.get_number_of_edges(), "src", src, flush=True)
run("fwd one counter per bin (round 3)", gr.forward, {"GRX_BIN_SUB": 1})
run("fwd four sub-counters per bin", gr.forward, {"GRX_BIN_SUB": 4})
run("fwd one counter per bin again", gr.forward, {"GRX_BIN_SUB": 1})
run("fwd four sub-counters again", gr.forward, {"GRX_BIN_SUB": 4})
run("fwd four sub-counters + pair stores", gr.forward, {"GRX_BIN_SUB": 4, "GRX_BIN_PAIR": 1})
if name != "lj":
    run("fwd four sub-counters, 32-bit entries", gr.forward, {"GRX_BIN_SUB": 4, "GRX_BIN_E16": 0})
    run("fwd one counter, 32-bit entries", gr.forward, {"GRX_BIN_SUB": 1, "GRX_BIN_E16": 0})
