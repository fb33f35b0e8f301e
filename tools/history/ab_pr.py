"""PageRank layouts on a stand-in: plain transpose pull (0x40) against the XCD-blocked layout (0x80).
    python tools/ab_pr.py lj|kron"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
props, csr, _, _ = bench.load_workload(gr, name)
ctx = gr.multi_context_t(0)
V = csr.number_of_rows
p = torch.empty(V, dtype=torch.float32, device="cuda")
res = gr.pr_result_t(p)
out = {}
for label, flags in (("plain transpose pull", 0x40), ("XCD-blocked layout", 0x80), ("engine's choice", 0)):
    G = gr.build_graph(props, csr, ctx)
    par = gr.pr_param_t(0.85, 1e-6, gr.options_t(engine_flags=flags))
    t0 = time.perf_counter()
    gr.pr_run(G, par, res, ctx)
    ctx.synchronize()
    first = (time.perf_counter() - t0) * 1e3
    ts = sorted(gr.pr_run(G, par, res, ctx) for _ in range(5))
    out[label] = p.cpu().numpy().copy()
    print("%-24s %.3f ms (%d iterations, %.4f ms each), first call %.1f ms" % (label, ts[2], res.iterations, ts[2] / res.iterations, first), flush=True)
    del G
a, b = out["plain transpose pull"], out["XCD-blocked layout"]
print("max |plain - blocked| = %.3e" % float(np.abs(a.astype(np.float64) - b).max()))
