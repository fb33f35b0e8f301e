"""A/B runs of the BFS engine's tuning knobs in ONE process (one graph build):
    python tools/ab_bfs.py [lj|kron] [group ...]      groups: do td knobs bin bin2 claim
Every configuration is checked against the first one's depths.  Prints one line per
configuration: wall ms per BFS (reset + enact, median), enact ms (events), per-level profile."""
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
groups = sys.argv[2:] or ["do", "td", "knobs"]
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
V = G.get_number_of_vertices()
d = torch.empty(V, dtype=torch.int32, device="cuda")
KNOBS = ("GRX_BIN_CLAIM", "GRX_MID_VERSION", "GRX_BIN_MAX_DEGREE", "GRX_TD_BITMAP", "GRX_TD_PRE", "GRX_TD_BIN", "GRX_BIN_MIN_EDGES", "GRX_LEVEL_MINW", "GRX_BU_BATCH", "GRX_DO_ALPHA", "GRX_DO_BETA", "GRX_DO_BACK_DIV", "GRX_LEVEL_WG_PER_CU", "GRX_PACE_DEPTH")
ref = None


def run(label, direction, variant=0, env=None, reps=15, profile=True):
    global ref
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                     advance_direction=direction, engine_flags=(variant << 8))
    for _ in range(3):
        gr.bfs(G, src, d, None, ctx, o)
    torch.cuda.synchronize()
    ctx.synchronize()
    walls, enacts = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        ms = gr.bfs(G, src, d, None, ctx, o)
        ctx.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
        enacts.append(ms)
    st = gr.run_stats(ctx)
    h = d.cpu().numpy()
    if ref is None:
        ref = h.copy()
    same = bool(np.array_equal(h, ref))
    lv = ""
    if profile:
        po = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                          advance_direction=direction, engine_flags=(variant << 8) | gr.FLAG_PROFILE)
        best = None
        for _ in range(2):
            gr.bfs(G, src, d, None, ctx, po)
            prof = gr.level_profile(ctx)
            t = sum(l["advance_ms"] for l in prof)
            if best is None or t < best[0]:
                best = (t, prof)
        lv = " ".join("%d/%d:%s%.0f+h%.0f" % (l["frontier_size"], l["edges"], {0: "T", 1: "B", 2: "N", 3: "M"}.get(l.get("bottom_up"), "?"),
                                              l["advance_ms"] * 1e3, l["other_ms"] * 1e3) for l in best[1])
    walls.sort()
    enacts.sort()
    print("%-34s wall med %.3f min %.3f | enact med %.3f min %.3f | GTEPS %.1f | same %s | groups %s | %s"
          % (label, walls[len(walls) // 2], walls[0], enacts[len(enacts) // 2], enacts[0],
             st["edges_visited"] / (enacts[len(enacts) // 2] * 1e6), same, st["aux"], lv), flush=True)


print("workload", name, "V", V, "E", G.get_number_of_edges(), "src", src, flush=True)
if "do" in groups:
    run("DO default", gr.optimized)
    for b in (2, 4):
        run("DO batch%d" % b, gr.optimized, env={"GRX_BU_BATCH": b})
if "knobs" in groups:
    for bd in (32, 64, 128):
        run("DO back_div %d" % bd, gr.optimized, env={"GRX_DO_BACK_DIV": bd})
    for be in (8, 12):
        run("DO beta %d" % be, gr.optimized, env={"GRX_DO_BETA": be})
    for pd in (1, 3):
        run("DO pace %d" % pd, gr.optimized, env={"GRX_PACE_DEPTH": pd}, profile=False)
if "td" in groups:
    run("TD main path (v0)", gr.forward)
    run("TD v7 (v0, plan+advance kernels)", gr.forward, variant=7)
if "bin" in groups:
    # forward-only run: round-1 body, bitmap pre-filter alone, binned fat levels (with / without the pre-filter
    # on the thin levels), binning thresholds
    run("TD round-1 body (no bitmap)", gr.forward, env={"GRX_TD_BITMAP": 0})
    run("TD bitmap pre-filter, no bins", gr.forward, env={"GRX_TD_BIN": 0})
    run("TD bins (default 1M) + pre-filter", gr.forward)
    run("TD bins, no pre-filter", gr.forward, env={"GRX_TD_PRE": 0})
    for m in (1 << 17, 1 << 18, 1 << 22):
        run("TD bins min_edges %d" % m, gr.forward, env={"GRX_BIN_MIN_EDGES": m})
    run("DO default (reference point)", gr.optimized)
if "bin2" in groups:
    run("TD round-1 body (no bitmap)", gr.forward, env={"GRX_TD_BITMAP": 0})
    run("TD bins, default rule", gr.forward)
    run("TD bins, every fat level", gr.forward, env={"GRX_BIN_MAX_DEGREE": 0})
if "claim" in groups:
    # claim phase of the binned levels: slices claimed in the owning XCD's L2 (round-2 first version) vs the sweep
    run("TD bins, slice claim (v2)", gr.forward, env={"GRX_BIN_CLAIM": 2})
    run("TD bins, sweep claim (v3)", gr.forward, env={"GRX_BIN_CLAIM": 3})
    run("TD bins, sweep, every fat level", gr.forward, env={"GRX_BIN_CLAIM": 3, "GRX_BIN_MAX_DEGREE": 0})
    run("TD bins, sweep, mid v1", gr.forward, env={"GRX_BIN_CLAIM": 3, "GRX_MID_VERSION": 1})
