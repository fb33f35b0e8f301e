"""K forward searches (configs[1]: merge-path advance + compact filter, ASYNC_RETURN) back to back on one stand-in -- the command a
rocprofv3 kernel trace / counter pass of the headline is taken on:  python tools/fwd_loop.py [lj|kron|twitter] [K] [fwd|do]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
direction = gr.optimized if (len(sys.argv) > 3 and sys.argv[3] == "do") else gr.forward
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
d = torch.empty(G.get_number_of_vertices(), dtype=torch.int32, device="cuda")
o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                 advance_direction=direction, engine_flags=gr.FLAG_ASYNC_RETURN)
for _ in range(K):
    gr.bfs(G, src, d, None, ctx, o)
ctx.synchronize()
print("done", gr.run_stats(ctx))
