"""Randomised parity sweep on the GPU (test infrastructure; calls the oracle):
    python tests/tools/fuzz_gpu.py [seconds, default 60] [seed]
Random R-MAT / symmetric R-MAT / lattice graphs of random size, random sources, BFS (forward and
direction-optimising, back to back without host syncs in between) and SSSP (unit and random
weights) against the oracle.  Prints one line per graph and a final verdict."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gunrock_amd as gr  # noqa: E402
import oracle_lib as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
ctx = gr.multi_context_t(0)
t_end = time.time() + budget
n_graphs = n_checks = n_bad = 0
while time.time() < t_end:
    kind = ["rmat", "rmat_sym", "road"][int(rng.integers(0, 3))]
    if kind == "road":
        side = int(rng.integers(20, 700))
        V, E, a = side * side, 0, float(rng.uniform(0.45, 0.9))
        gseed = int(rng.integers(1, 1 << 30))
        props, c = gr.generate("road", V, 0, a, 0.0, 0.0, seed=gseed)
    else:
        V = int(rng.integers(50, 600_000))
        E = int(V * rng.uniform(0.5, 40.0))
        gseed = int(rng.integers(1, 1 << 30))
        props, c = gr.generate(kind, V, E, seed=gseed)
    V = len(c.row_offsets) - 1
    nnz = int(c.number_of_nonzeros)
    if nnz == 0:
        continue
    weighted = bool(rng.integers(0, 2))
    if weighted:
        c.nonzero_values = rng.integers(1, 200, nnz).astype(np.float32) / 4.0
        props.weighted = True
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    G = gr.build_graph(props, c, ctx)
    deg = np.diff(c.row_offsets)
    sources = [int(np.argmax(deg))] + [int(x) for x in rng.integers(0, V, 3)]
    d_i = torch.empty(V, dtype=torch.int32, device="cuda")
    d_f = torch.empty(V, dtype=torch.float32, device="cuda")
    bad = 0
    for s in sources:
        want_b, _ = O.bfs(g, s)
        for direction in (gr.forward, gr.optimized):
            o = gr.options_t(advance_direction=direction)
            gr.bfs(G, s, d_i, None, ctx, o)
            gr.bfs(G, s, d_i, None, ctx, o)  # back to back: the second starts while the first's tail drains
            n_checks += 1
            if not np.array_equal(d_i.cpu().numpy(), want_b):
                bad += 1
                got = d_i.cpu().numpy()
                w = np.nonzero(got != want_b)[0]
                print("  BFS MISMATCH", kind, V, nnz, "gseed", gseed, "src", s, "dir", direction, "n_wrong", len(w),
                      "first", [(int(i), int(got[i]), int(want_b[i])) for i in w[:6]], flush=True)
        want_s, _ = O.sssp(g, s)
        gr.sssp(G, s, d_f, None, ctx, gr.options_t())
        n_checks += 1
        if not np.array_equal(d_f.cpu().numpy(), want_s):
            bad += 1
            print("  SSSP MISMATCH", kind, V, nnz, "src", s, "weighted", weighted, flush=True)
    n_graphs += 1
    n_bad += bad
    print("%-8s V %7d E %9d weighted %d  %s" % (kind, V, nnz, weighted, "ok" if not bad else "BAD x%d" % bad), flush=True)
    del G
print("FUZZ %s: %d graphs, %d checks, %d mismatches" % ("PASSED" if n_bad == 0 else "FAILED", n_graphs, n_checks, n_bad))
sys.exit(1 if n_bad else 0)
