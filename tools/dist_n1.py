"""The partitioned (multi-GPU) BFS path at ONE rank on the bench workload: what the level-group machinery costs next
to the single-GPU enactor.
    python tools/dist_n1.py [lj|kron] [runs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrock_amd as gr  # noqa: E402
from gunrock_amd import distributed as D  # noqa: E402
from bench import WORKLOADS  # noqa: E402

wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "lj"]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
V = len(csr.row_offsets) - 1
E = int(csr.row_offsets[-1])
_, cin = gr.generate_rows(wl["kind"], wl["V"], wl["entries"], 0, V, wl["a"], wl["b"], wl["c"], seed=42, in_rows=True) \
    if wl["kind"] == "rmat" else (None, None)
src = int(np.argmax(np.diff(csr.row_offsets)))
eng = D.GrxEngine(props, csr, 0, 1, "cuda:0", E, in_rows=cin)
if os.environ.get("GRX_DIST_RCCL") == "1":
    eng.enable_library_transport(None)
d = eng.new_labels()
for _ in range(3):
    st = D.bfs(eng, None, src, d, optimized=True)
torch.cuda.synchronize()
ts = []
for _ in range(runs):
    t0 = time.perf_counter()
    st = D.bfs(eng, None, src, d, optimized=True)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("partitioned path, 1 rank, %s: wall per BFS med %.3f min %.3f ms | depth %d | edges %d | %.1f GTEPS | transport: %s"
      % (sys.argv[1] if len(sys.argv) > 1 else "lj", ts[len(ts) // 2], ts[0], st["search_depth"], st["edges_visited"],
         st["edges_visited"] / (ts[len(ts) // 2] * 1e6), eng.transport_description()))
