"""A/B of the bottom-up levels of the direction-optimising BFS in ONE process per graph:
    python tools/ab_bu.py [lj|kron|twitter] [reps]
first version (grx_bfs_kernels.hpp bfs_bottomup_block) vs second (bfs_bottomup2_block: one round trip per round of chunks,
unsettled lanes deferred), the second at several residencies.  Every configuration is checked against the depths of a
forward-only search.  One line per configuration: wall ms per BFS (reset + enact), enact ms, per-level profile (us)."""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
V = G.get_number_of_vertices()
d = torch.empty(V, dtype=torch.int32, device="cuda")
KNOBS = ("GRX_BU2", "GRX_BU_HEADS", "GRX_LEVEL_WG_PER_CU", "GRX_BU_BATCH", "GRX_PACE_DEPTH", "GRX_SEED_IN_RESET", "GRX_SOURCE_WG_PER_CU")
ref = None


def run(label, direction, env=None, sources=(src,)):
    global ref
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    o = gr.options_t(advance_direction=direction, engine_flags=gr.FLAG_ASYNC_RETURN)
    for _ in range(3):
        gr.bfs(G, src, d, None, ctx, o)
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
    po = gr.options_t(advance_direction=direction, engine_flags=gr.FLAG_PROFILE)
    best = None
    for _ in range(3):
        gr.bfs(G, src, d, None, ctx, po)
        prof = gr.level_profile(ctx)
        t = sum(l["advance_ms"] for l in prof)
        if best is None or t < best[0]:
            best = (t, prof)
    same = same and bool(np.array_equal(d.cpu().numpy(), ref))
    lv = " ".join("%d/%d:%s%.0f+h%.0f" % (l["frontier_size"], l["edges"], {0: "T", 1: "B", 2: "N", 3: "M"}.get(l.get("bottom_up"), "?"),
                                          l["advance_ms"] * 1e3, l["other_ms"] * 1e3) for l in best[1])
    print("%-44s step %.4f ms | enact %.4f | GTEPS %.1f | same %s | %s"
          % (label, step, st["elapsed_ms"], st["edges_visited"] / (step * 1e6), same, lv), flush=True)


print("workload", name, "V", V, "E", G.get_number_of_edges(), "src", src, flush=True)
run("forward (reference depths)", gr.forward, {})
run("DO first bottom-up body", gr.optimized, {"GRX_BU2": 0})
run("DO second body (default)", gr.optimized, {})
run("DO second body, 5 workgroups per CU", gr.optimized, {"GRX_LEVEL_WG_PER_CU": 5})
run("DO second body, reset and seed as two launches", gr.optimized, {"GRX_SEED_IN_RESET": 0})
run("DO second body, source level on 1 workgroup per CU", gr.optimized, {"GRX_SOURCE_WG_PER_CU": 1})
run("DO second body, source level on 2 workgroups per CU", gr.optimized, {"GRX_SOURCE_WG_PER_CU": 2})
run("DO second body, source level on 8 workgroups per CU", gr.optimized, {"GRX_SOURCE_WG_PER_CU": 8})
run("DO first body again", gr.optimized, {"GRX_BU2": 0})
run("DO second body again", gr.optimized, {})
# other sources: depths against the forward search from the same source
rng = np.random.default_rng(7)
deg = np.diff(csr.row_offsets)
bad = 0
for s2 in rng.choice(np.nonzero(deg > 0)[0], size=6, replace=False).tolist():
    outs = []
    for direction, env in ((gr.forward, {}), (gr.optimized, {}), (gr.optimized, {"GRX_BU2": 0})):
        for k in KNOBS:
            os.environ.pop(k, None)
        for k, v in env.items():
            os.environ[k] = str(v)
        gr.bfs(G, int(s2), d, None, ctx, gr.options_t(advance_direction=direction))
        outs.append(d.cpu().numpy().copy())
    ok = np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    bad += 0 if ok else 1
    print("source %d (out-degree %d): second body == first body == forward: %s" % (s2, deg[s2], ok), flush=True)
print("mismatching sources:", bad)
