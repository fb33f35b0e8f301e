"""Near-far SSSP bucket-width sweep on the weighted road stand-in (one graph build):
    python tools/ab_sssp_delta.py [scale ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

wl = WORKLOADS["road"]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], 1.0, seed=42)
src = (4894 // 2) * 4894 + 4894 // 2
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
V = G.get_number_of_vertices()
d = torch.empty(V, dtype=torch.float32, device="cuda")
ref = None
for sc in [float(x) for x in (sys.argv[1:] or ["1", "0.25", "0.5", "2"])]:
    os.environ["GRX_NF_DELTA_SCALE"] = str(sc)
    gr.sssp(G, src, d, None, ctx, gr.options_t())
    ts = sorted(gr.sssp(G, src, d, None, ctx, gr.options_t()) for _ in range(3))
    st = gr.run_stats(ctx)
    h = d.cpu().numpy()
    if ref is None:
        ref = h.copy()
    print("delta x%-4g enact med %.1f ms  iterations %d  relaxations %d  phases %s  same %s"
          % (sc, ts[1], st["search_depth"], st["edges_visited"], st["aux"], bool(np.array_equal(h, ref))), flush=True)
