"""Single-rank run of the partitioned BFS path (P = 1: the exchanges are device copies):
    python tools/run_dist1.py lj|kron|small [runs] [overlap 0|1] [optimized 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gunrock_amd as gr
from gunrock_amd import distributed as D
from bench import WORKLOADS
wl = WORKLOADS[sys.argv[1]]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
overlap = len(sys.argv) > 3 and sys.argv[3] == "1"
optimized = not (len(sys.argv) > 4 and sys.argv[4] == "0")
V = wl["V"]
props, c = gr.generate(wl["kind"], V, wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
cin = None
if wl["kind"] == "rmat":
    _, cin = gr.generate_rows(wl["kind"], V, wl["entries"], 0, V, wl["a"], wl["b"], wl["c"], seed=42, in_rows=True)
src = int(np.argmax(np.diff(c.row_offsets)))
eng = D.GrxEngine(props, c, 0, 1, "cuda:0", int(c.number_of_nonzeros), in_rows=cin, overlap=overlap)
d = torch.empty(V, dtype=torch.int32, device="cuda:0")
ts = []
for _ in range(runs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = D.bfs(eng, None, src, d, optimized=optimized)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("dist P=1", sys.argv[1], "overlap", overlap, "optimized", optimized, "wall ms", [round(t, 3) for t in ts], st,
      "GTEPS", round(st["edges_visited"] / (min(ts) * 1e6), 1))
