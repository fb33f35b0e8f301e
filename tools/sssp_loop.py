"""K weighted SSSP runs (U{1..1000} weights, the bench's sssp_<graph>_w section) back to back on one stand-in -- the command a
rocprofv3 kernel trace is taken on:  python tools/sssp_loop.py [lj|kron] [K]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS, pair_hash_weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
w = pair_hash_weights(csr)
csr_w = gr.csr_t.from_arrays(csr.row_offsets, csr.column_indices, w)
G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=True, symmetric=False), csr_w, ctx)
d = torch.empty(csr.number_of_rows, dtype=torch.float32, device="cuda")
for _ in range(K):
    gr.sssp(G, src, d, None, ctx, gr.options_t())
ctx.synchronize()
print("done", gr.run_stats(ctx))
