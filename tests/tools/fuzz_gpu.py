"""Randomised parity sweep on the GPU (test infrastructure; calls the oracle):
    python tests/tools/fuzz_gpu.py [seconds, default 60] [seed]
Random R-MAT / symmetric R-MAT / lattice graphs of random size, random sources, BFS (forward and
direction-optimising, synchronous and with GRX_FLAG_ASYNC_RETURN back to back without host syncs in
between; the forward runs alternately with every level forced through the binned scatter + claim
kernels) and SSSP (unit and random weights) against the oracle.  Prints one line per graph and a
final verdict.  tests/test_fuzz_gpu.py runs a bounded slice of the same sweep under pytest -m gpu."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sweep(budget_s, seed=1234, log=print, max_vertices=600_000):
    """-> (graphs, checks, mismatches).  Every mismatch is logged with what is needed to replay it."""
    import torch
    import gunrock_amd as gr
    import oracle_lib as O
    rng = np.random.default_rng(seed)
    ctx = gr.multi_context_t(0)
    t_end = time.time() + budget_s
    n_graphs = n_checks = n_bad = 0
    saved_env = os.environ.get("GRX_BIN_MIN_EDGES")
    try:
        while time.time() < t_end:
            kind = ["rmat", "rmat_sym", "road"][int(rng.integers(0, 3))]
            if kind == "road":
                side = int(rng.integers(20, max(21, int(max_vertices ** 0.5))))
                V, a = side * side, float(rng.uniform(0.45, 0.9))
                gseed = int(rng.integers(1, 1 << 30))
                props, c = gr.generate("road", V, 0, a, 0.0, 0.0, seed=gseed)
            else:
                V = int(rng.integers(50, max_vertices))
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
            # sparse graphs: every other one small enough to stay on the level-synchronous kernels takes the
            # block-asynchronous path anyway (grx_block.hip; the threshold is read when the handle first searches)
            os.environ["GRX_BLOCK_MIN_V"] = "512" if int(rng.integers(0, 2)) else "65536"
            os.environ["GRX_BLOCK"] = "1"
            G = gr.build_graph(props, c, ctx)
            deg = np.diff(c.row_offsets)
            sources = [int(np.argmax(deg))] + [int(x) for x in rng.integers(0, V, 3)]
            d_i = torch.empty(V, dtype=torch.int32, device="cuda")
            d_f = torch.empty(V, dtype=torch.float32, device="cuda")
            bad = 0
            for si, s in enumerate(sources):
                want_b, _ = O.bfs(g, s)
                # forward runs: every other source with ALL levels binned (scatter + claim kernels), else the default threshold
                force_bins = si % 2 == 0
                if force_bins:
                    os.environ["GRX_BIN_MIN_EDGES"] = "1"
                    os.environ["GRX_BIN_MAX_DEGREE"] = "0"
                else:
                    os.environ.pop("GRX_BIN_MIN_EDGES", None)
                    os.environ.pop("GRX_BIN_MAX_DEGREE", None)
                for direction in (gr.forward, gr.optimized):
                    flags = gr.FLAG_ASYNC_RETURN if rng.integers(0, 2) else 0
                    o = gr.options_t(advance_direction=direction, engine_flags=flags)
                    gr.bfs(G, s, d_i, None, ctx, o)
                    gr.bfs(G, s, d_i, None, ctx, o)  # back to back: with ASYNC_RETURN the second starts while the first's tail drains
                    n_checks += 1
                    got = d_i.cpu().numpy()
                    if not np.array_equal(got, want_b):
                        bad += 1
                        w = np.nonzero(got != want_b)[0]
                        log("  BFS MISMATCH %s V %d E %d gseed %d src %d dir %d flags %d bins_forced %d n_wrong %d first %s"
                            % (kind, V, nnz, gseed, s, direction, flags, force_bins, len(w),
                               [(int(i), int(got[i]), int(want_b[i])) for i in w[:6]]))
                want_s, _ = O.sssp(g, s)
                # weighted dense graphs: the binned relaxation (grx_relax.hpp) on every level the head kernel plans, or on the
                # fat ones only; every other time with so many parts that the bins of a level are relaxed by racing workgroups
                os.environ["GRX_RBIN_MIN_GRAPH_EDGES"] = "0"
                os.environ["GRX_RBIN_MIN_EDGES"] = "1" if force_bins else "50000"
                if int(rng.integers(0, 2)):
                    os.environ["GRX_RBIN_PARTS"] = "4096"
                else:
                    os.environ.pop("GRX_RBIN_PARTS", None)
                gr.sssp(G, s, d_f, None, ctx, gr.options_t())
                n_checks += 1
                if not np.array_equal(d_f.cpu().numpy(), want_s):
                    bad += 1
                    log("  SSSP MISMATCH %s V %d E %d gseed %d src %d weighted %d" % (kind, V, nnz, gseed, s, weighted))
            n_graphs += 1
            n_bad += bad
            log("%-8s V %7d E %9d weighted %d  %s" % (kind, V, nnz, weighted, "ok" if not bad else "BAD x%d" % bad))
            del G
    finally:
        os.environ.pop("GRX_BLOCK_MIN_V", None)
        os.environ.pop("GRX_BLOCK", None)
        os.environ.pop("GRX_BIN_MAX_DEGREE", None)
        for k in ("GRX_RBIN_MIN_GRAPH_EDGES", "GRX_RBIN_MIN_EDGES", "GRX_RBIN_PARTS"):
            os.environ.pop(k, None)
        if saved_env is None:
            os.environ.pop("GRX_BIN_MIN_EDGES", None)
        else:
            os.environ["GRX_BIN_MIN_EDGES"] = saved_env
    return n_graphs, n_checks, n_bad


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
    ng, nc, nb = sweep(budget, seed, log=lambda m: print(m, flush=True))
    print("FUZZ %s: %d graphs, %d checks, %d mismatches" % ("PASSED" if nb == 0 else "FAILED", ng, nc, nb))
    sys.exit(1 if nb else 0)
