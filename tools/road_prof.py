"""Per launch group of ONE profiled SSSP on the road stand-in (GRX_FLAG_PROFILE: a sync after every group): which body ran, how
many levels it advanced, vertices / edges, kernel times -- where the ~90 ms of the weighted search go.
    python tools/road_prof.py [w|unit]      ROAD_AB_SIDE=<n> for a smaller lattice"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

weighted = (sys.argv[1] if len(sys.argv) > 1 else "w") == "w"
wl = WORKLOADS["road"]
side = int(os.environ.get("ROAD_AB_SIDE", "0")) or int(round(wl["V"] ** 0.5))
V = side * side
src = (side // 2) * side + side // 2
ctx = gr.multi_context_t(0)
props, csr = gr.generate("road", V, 0, wl["a"], wl["b"], 1.0 if weighted else wl["c"], seed=42)
G = gr.build_graph(props, csr, ctx, device="cuda:0")
d = torch.empty(V, dtype=torch.float32, device="cuda:0")
flags = gr.FLAG_PROFILE | (0 if weighted else gr.FLAG_SSSP_NO_BFS)
po = gr.options_t(advance_load_balance=gr.merge_path, engine_flags=flags)
gr.sssp(G, src, d, None, ctx, gr.options_t())
gr.sssp(G, src, d, None, ctx, po)
prof = gr.level_profile(ctx, capacity=65536)
prev = -1
rows = []
for l in prof:
    lv = l["bu_open"]
    rows.append((l["bu_probes"], l["bottom_up"], lv - prev, l["frontier_size"], l["edges"], l["advance_ms"], l["other_ms"]))
    prev = lv
rows = np.array(rows, dtype=np.float64)
print("groups %d, levels %d, level-kernel ms %.2f, head ms %.2f" % (len(rows), prev + 1, rows[:, 5].sum(), rows[:, 6].sum()))
for mode in (0, 2, 3):
    for split in (0, 2):
        m = (rows[:, 0] == mode) & (rows[:, 1] == split)
        if not m.any():
            continue
        r = rows[m]
        print("  mode %d%s: groups %5d  levels %6d  vertices %9d  edges %10d  level ms %7.2f (%.1f us / group, %.2f us / level)  head ms %6.2f"
              % (mode, " (bucket pull)" if split else "", len(r), r[:, 2].sum(), r[:, 3].sum(), r[:, 4].sum(), r[:, 5].sum(),
                 1e3 * r[:, 5].sum() / len(r), 1e3 * r[:, 5].sum() / max(1, r[:, 2].sum()), r[:, 6].sum()))
# frontier size per level (vertices of a group / its levels) against time per level
m = rows[:, 2] > 0
per_level_v = rows[m, 3] / rows[m, 2]
per_level_us = 1e3 * rows[m, 5] / rows[m, 2]
for lo, hi in ((0, 1024), (1024, 4096), (4096, 8192), (8192, 16384), (16384, 32768), (32768, 65536), (65536, 1 << 30)):
    k = (per_level_v >= lo) & (per_level_v < hi)
    if k.any():
        print("  groups with %6d <= vertices / level < %6d: %5d groups %6d levels  %.2f us / level   %s"
              % (lo, hi, k.sum(), rows[m][k, 2].sum(), (per_level_us[k] * rows[m][k, 2]).sum() / rows[m][k, 2].sum(),
                 "modes " + str(sorted(set(rows[m][k, 0].astype(int))))))
