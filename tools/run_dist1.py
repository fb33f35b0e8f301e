"""The partitioned (multi-GPU) BFS path with ONE rank on one GPU: what a level group costs without
the network.    python tools/run_dist1.py [lj|kron] [runs]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from gunrock_amd import distributed as D  # noqa: E402
from bench import WORKLOADS  # noqa: E402

wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "lj"]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
V, entries = wl["V"], wl["entries"]
props, mine = gr.generate_rows(wl["kind"], V, entries, 0, V, wl["a"], wl["b"], wl["c"], seed=42)
mine_in = None
if wl["kind"] == "rmat":
    _, mine_in = gr.generate_rows(wl["kind"], V, entries, 0, V, wl["a"], wl["b"], wl["c"], seed=42, in_rows=True)
src = int(np.argmax(np.diff(mine.row_offsets)))
eng = D.GrxEngine(props, mine, 0, 1, "cuda:0", int(mine.number_of_nonzeros), in_rows=mine_in)
dist_t = torch.empty(V, dtype=torch.int32, device="cuda:0")
for _ in range(3):
    st = D.bfs(eng, None, src, dist_t)
torch.cuda.synchronize()
ts = []
for _ in range(runs):
    t0 = time.perf_counter()
    st = D.bfs(eng, None, src, dist_t)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("partitioned path, 1 rank: wall med %.3f ms min %.3f | enact %.3f ms | depth %d edges %d"
      % (ts[len(ts) // 2], ts[0], st["elapsed_ms"], st["search_depth"], st["edges_visited"]))
