"""Same-hardware baseline: time the REFERENCE's own hipified GPU path
(oracle/_ref/libgunrock_ref_gpu.so, compiled from /root/reference sources) on the
bench workloads.  Test infrastructure / reporting only (BASELINE.md "Ref-GPU").

    python oracle/ref_gpu_bench.py [lj|small|kron|road] [runs]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gunrock_amd as gr  # noqa: E402  (generator only)
import oracle_lib as O  # noqa: E402
from bench import WORKLOADS  # noqa: E402


def main():
    wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "lj"]
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
    g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
    src = int(np.argmax(np.diff(g.row_offsets)))
    L = O.ref_gpu()
    h = L.ref_gpu_graph_create(g.n_vertices, g.n_edges, g.row_offsets, g.column_indices, g.values)
    want, _, ev = O.bfs_queue(g, src)
    out = {"workload": wl["name"], "source": src, "edges_visited": ev}
    d = np.empty(g.n_vertices, np.int32)
    for name, lb, flt, alg in (("block_mapped", 2, 0, 1), ("merge_path", 4, 0, 1),
                               ("merge_path+predicated", 4, 1, 1), ("thread_mapped", 0, 0, 1)):
        times = []
        for _ in range(runs):
            ms = L.ref_gpu_bfs(h, src, lb, flt, alg, d)
            times.append(ms)
        ok = bool(np.array_equal(d, want))
        best = min(times)
        out["bfs_" + name] = {"ms_min": round(best, 3), "ms_all": [round(t, 3) for t in times],
                              "mteps": round(ev / (best * 1e3), 1) if best > 0 else None, "matches_oracle": ok}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
