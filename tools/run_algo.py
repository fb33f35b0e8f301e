"""Tiny workload driver for profiling: run one algorithm a few times on a bench workload.
    python tools/run_algo.py bfs|sssp|ssspu|pr lj|kron|road|small [runs] [engine_flags] [lb] [direction: forward|optimized]
sssp: weights U{1..1000} (drawn here for pattern graphs); ssspu: the graph's own weights (1.0 for pattern files)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

algo = sys.argv[1]
wl = WORKLOADS[sys.argv[2]]
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
lb = getattr(gr, sys.argv[5]) if len(sys.argv) > 5 else gr.merge_path
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
if algo == "sssp" and sys.argv[2] == "road":
    props, csr = gr.generate("road", wl["V"], 0, wl["a"], 0.0, 1.0, seed=42)  # the weighted road variant of bench.py
elif algo == "sssp" and not props.weighted:
    from bench import pair_hash_weights  # the weights of bench.py's sssp_{lj,kron}_w sections: same workload
    csr.nonzero_values = pair_hash_weights(csr)
    csr._device = None
    props.weighted = True
src = int(np.argmax(np.diff(csr.row_offsets)))
if sys.argv[2] == "road":
    src = (4894 // 2) * 4894 + 4894 // 2
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
V = G.get_number_of_vertices()
o = gr.options_t(advance_load_balance=lb, engine_flags=flags)
if len(sys.argv) > 6:
    o.advance_direction = getattr(gr, sys.argv[6])
times = []
for _ in range(runs):
    if algo == "bfs":
        d = torch.empty(V, dtype=torch.int32, device="cuda")
        times.append(gr.bfs(G, src, d, None, ctx, o))
    elif algo in ("sssp", "ssspu"):
        d = torch.empty(V, dtype=torch.float32, device="cuda")
        times.append(gr.sssp(G, src, d, None, ctx, o))
    else:
        p = torch.empty(V, dtype=torch.float32, device="cuda")
        res = gr.pr_result_t(p)
        times.append(gr.pr_run(G, gr.pr_param_t(0.85, 1e-6, o), res, ctx))
st = gr.run_stats(ctx)
print("algo", algo, "ms", [round(t, 3) for t in times], "stats", st,
      "mteps", round(st["edges_visited"] / (min(times) * 1e3), 1))
