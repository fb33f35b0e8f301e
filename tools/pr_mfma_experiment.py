"""The north_star's MFMA experiment for the PageRank update, measured ("treat as an experiment").

    python tools/pr_mfma_experiment.py [K ...]      default K = 2048 4096 8192

PageRank's pull is y = A^T x with ONE right-hand side: a matrix-vector product.  The only sub-problem of it that
is GEMM-shaped at all is the dense-ish HUB x HUB block of a scale-free graph: the in-edges of the top-K rows (by
in-degree) that come from the top-K sources (by out-degree).  This script
  1. takes the C4' kron stand-in, ranks vertices by degree and counts how many of the E edges fall inside the
     K x K hub block (its density is what an MFMA formulation lives on);
  2. times y_h = B x_h for that block as a dense fp32 matrix on the matrix cores -- torch.mv / torch.mm through
     hipBLASLt, i.e. the vendor's f32-input MFMA kernels (v_mfma_f32_32x32x2_f32): the ceiling a hand-written
     tile of the same shape could reach -- with 1 and with 32 right-hand sides (an MFMA tile needs N >= 16-32
     columns to be fed; PageRank has one, so 31/32 of the tile would multiply padding);
  3. sets that against what the engine's sparse pull spends on the SAME edges (its measured time per gathered
     edge x the edges inside the block) and against the bytes each form moves.
fp32 is required: ranks are ~1e-7 .. 1e-3 and the tolerance is 1e-6 absolute; bf16 / f16 inputs (the fast MFMA
rates) carry 2^-8 .. 2^-11 relative error per product and cannot meet it.
Prints one JSON object; DESIGN.md section 3.4 (HISTORY.md section 3.4) quotes it."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

Ks = [int(x) for x in sys.argv[1:]] or [2048, 4096, 8192]
wl = WORKLOADS["kron"]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
V, E = csr.number_of_rows, csr.number_of_nonzeros
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
p = torch.empty(V, dtype=torch.float32, device="cuda")
res = gr.pr_result_t(p)
par = gr.pr_param_t(0.85, 1e-6)
gr.pr_run(G, par, res, ctx)
ts = [gr.pr_run(G, par, res, ctx) for _ in range(5)]
ms_iter = sorted(ts)[2] / res.iterations
ns_per_edge = ms_iter * 1e6 / E
deg = np.diff(csr.row_offsets).astype(np.int64)      # symmetric graph: out-degree == in-degree
order = np.argsort(-deg, kind="stable")
rank = np.empty(V, dtype=np.int64)
rank[order] = np.arange(V)
src = np.repeat(np.arange(V, dtype=np.int64), deg)
dst = csr.column_indices.astype(np.int64)
rs, rd = rank[src], rank[dst]
out = {"workload": wl["name"], "V": V, "E": E, "engine_pull_ms_per_iteration": round(ms_iter, 4),
       "engine_ns_per_gathered_edge": round(ns_per_edge, 5), "blocks": []}


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


for K in Ks:
    inside = (rs < K) & (rd < K)
    n_in = int(inside.sum())
    # multiplicities: the loader keeps duplicate edges
    B = np.bincount(rd[inside] * K + rs[inside], minlength=K * K).reshape(K, K).astype(np.float32)
    nnz_cells = int((B != 0).sum())
    Bd = torch.from_numpy(B).cuda()
    x1 = torch.rand(K, dtype=torch.float32, device="cuda")
    x32 = torch.rand(K, 32, dtype=torch.float32, device="cuda")
    t_mv = timed(lambda: torch.mv(Bd, x1))
    t_mm = timed(lambda: torch.mm(Bd, x32))
    sparse_ms = n_in * ns_per_edge * 1e-6
    out["blocks"].append({
        "K": K, "edges_inside": n_in, "share_of_E": round(n_in / E, 4), "distinct_cells": nnz_cells,
        "cell_density": round(nnz_cells / (K * K), 4),
        "dense_f32_bytes": 4 * K * K, "sparse_bytes_8_per_edge": 8 * n_in,
        "dense_mv_ms_1_rhs": round(t_mv, 4), "dense_mm_ms_32_rhs": round(t_mm, 4),
        "dense_mm_tflops_32_rhs": round(2.0 * K * K * 32 / (t_mm * 1e-3) / 1e12, 2),
        "engine_sparse_ms_same_edges": round(sparse_ms, 4),
        "dense_over_sparse_time_1_rhs": round(t_mv / max(sparse_ms, 1e-9), 2)})
print(json.dumps(out))
