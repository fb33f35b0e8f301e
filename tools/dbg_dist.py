import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gunrock_amd as gr
from gunrock_amd import distributed as D
import oracle_lib as O
V, E = 1 << 16, 1 << 20
props, c = gr.generate("rmat", V, E, seed=2)
_, cin = gr.generate_rows("rmat", V, E, 0, V, seed=2, in_rows=True)
g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
src = int(np.argmax(np.diff(g.row_offsets)))
want, _, ev = O.bfs_queue(g, src)
for overlap in (False, True):
    eng = D.GrxEngine(props, c, 0, 1, "cuda:0", E, in_rows=cin, overlap=overlap)
    d = torch.empty(V, dtype=torch.int32, device="cuda:0")
    for optimized in (False, True):
        st = D.bfs(eng, None, src, d, optimized=optimized)
        got = d.cpu().numpy()
        bad = np.flatnonzero(got != want)
        print("overlap", overlap, "opt", optimized, "mismatch", len(bad), "stats", st, "ev", ev, "want depth", want[want != 2**31-1].max())
        if len(bad): print("  first bad", bad[:8], got[bad[:8]], want[bad[:8]])
