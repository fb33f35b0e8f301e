"""Round-3 A/B of the binned forward levels in ONE process per graph (one graph build):
    python tools/ab_r3.py [lj|kron|twitter] [reps]
scatter: first version (inside the level kernel, 256-thread workgroups) vs second (bfs_scatter2_kernel, 1024 threads);
sweep claim: first version (1024 threads, one workgroup per CU) vs second (512 threads, four per CU).
Every configuration is checked against the first one's depths.  One line per configuration: wall ms per BFS
(reset + enact, median / min), enact ms, per-level profile (level kernels + head kernel, us)."""
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
KNOBS = ("GRX_BU2", "GRX_BIN_HINT", "GRX_BU_HEADS", "GRX_BIN_E16", "GRX_BIN_SCATTER", "GRX_BIN_SWEEP", "GRX_SW2_ITEMS", "GRX_SW2_WG_PER_CU", "GRX_SC2_WG_PER_CU", "GRX_BIN_MIN_EDGES")
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
run("fwd scatter1 sweep1 (round 2)", gr.forward, {"GRX_BIN_SCATTER": 1, "GRX_BIN_SWEEP": 1})
run("fwd scatter2 sweep1", gr.forward, {"GRX_BIN_SWEEP": 1, "GRX_BIN_E16": 0})
run("fwd scatter2 sweep3 32-bit entries", gr.forward, {"GRX_BIN_E16": 0})
run("fwd default (scatter2 sweep3 16-bit)", gr.forward, {})
run("fwd default again", gr.forward, {})
run("fwd, every group with scatter + sweep", gr.forward, {"GRX_BIN_HINT": 0})
run("DO default", gr.optimized, {})
run("DO first bottom-up body", gr.optimized, {"GRX_BU2": 0})
run("DO first body, no two-neighbour array", gr.optimized, {"GRX_BU_HEADS": 0})
run("DO default again", gr.optimized, {})
