"""Road stand-in (bench.py's sssp_road section: 4894 x 4894 lattice) A/B inside ONE process: SSSP with U{1..1000} weights
(near-far schedule) and / or unit weights (BFS engine) under a list of environment settings, each timed over K runs, CRC of
the distances printed per line (equal CRCs == equal results: the fixed point does not depend on the schedule).

    python tools/road_ab.py [w|unit|both] [K] "NAME=VAL,NAME=VAL" "NAME=VAL" ...      ("-" = defaults)
    ROAD_AB_SIDE=<n>: an n x n lattice instead (quick runs);  ROAD_AB_CHECK=1: oracle's fixed-point check of the first setting
"""
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "w"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
settings = sys.argv[3:] or ["-"]
wl = dict(WORKLOADS["road"])
side = int(os.environ.get("ROAD_AB_SIDE", "0")) or int(round(wl["V"] ** 0.5))
V = side * side
src = (side // 2) * side + side // 2
ctx = gr.multi_context_t(0)


def apply(setting):
    touched = []
    if setting != "-":
        for kv in setting.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
            touched.append(k)
    return touched


for weighted in ([True] if which == "w" else [False] if which == "unit" else [False, True]):
    t0 = time.time()
    props, csr = gr.generate("road", V, 0, wl["a"], wl["b"], 1.0 if weighted else wl["c"], seed=42)
    G = gr.build_graph(props, csr, ctx, device="cuda:0")
    d = torch.empty(V, dtype=torch.float32, device="cuda:0")
    print("== road %d x %d, %s weights: V %d E %d (setup %.1f s)" % (side, side, "U{1..1000}" if weighted else "unit", V,
                                                                    G.get_number_of_edges(), time.time() - t0), flush=True)
    o = gr.options_t(advance_load_balance=gr.merge_path)
    for si, setting in enumerate(settings):
        touched = apply(setting)
        gr.sssp(G, src, d, None, ctx, o)
        ctx.synchronize()
        ts = []
        for _ in range(K):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            gr.sssp(G, src, d, None, ctx, o)
            ctx.synchronize()
            ts.append((time.perf_counter() - t1) * 1e3)
        st = gr.run_stats(ctx)
        mine = d.cpu().numpy()
        crc = zlib.crc32(mine.tobytes()) & 0xffffffff
        print("  %-44s  ms %s  best %.2f | iterations %d relaxed %d phases %s | crc %08x" % (
            setting, " ".join("%.2f" % t for t in ts), min(ts), st["search_depth"], st["edges_visited"],
            st.get("reserved", st.get("aux", "-")), crc), flush=True)
        if si == 0 and os.environ.get("ROAD_AB_CHECK") == "1":
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
            print("    oracle fixed-point violations: %d" % O.check_sssp(g, src, mine), flush=True)
        for k in touched:
            del os.environ[k]
    del G, d, csr
