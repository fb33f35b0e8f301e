"""Weighted SSSP (U{1..1000}, bench.py's sssp_<graph>_w sections) on a dense stand-in under environment settings, one process, CRC of the
distances per line:      python tools/ab_ssspw.py [lj|kron] [K] "NAME=VAL,..." ...       ("-" = defaults)"""
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS, pair_hash_weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
settings = sys.argv[3:] or ["-"]
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
w = pair_hash_weights(csr)
csr_w = gr.csr_t.from_arrays(csr.row_offsets, csr.column_indices, w)
G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=True, symmetric=False), csr_w, ctx)
d = torch.empty(csr.number_of_rows, dtype=torch.float32, device="cuda")
print("workload", name, "V", csr.number_of_rows, "E", csr.number_of_nonzeros, flush=True)
for setting in settings:
    touched = []
    if setting != "-":
        for kv in setting.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
            touched.append(k)
    for _ in range(2):
        gr.sssp(G, src, d, None, ctx, gr.options_t())
    ctx.synchronize()
    ts = []
    for _ in range(K):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gr.sssp(G, src, d, None, ctx, gr.options_t())
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    st = gr.run_stats(ctx)
    print("  %-36s best %.3f ms median %.3f | iterations %d relaxed %d | crc %08x" % (
        setting, min(ts), sorted(ts)[len(ts) // 2], st["search_depth"], st["edges_visited"],
        zlib.crc32(d.cpu().numpy().tobytes()) & 0xffffffff), flush=True)
    for k in touched:
        del os.environ[k]
