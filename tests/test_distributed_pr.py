"""Partitioned PageRank (gunrock_amd/distributed.py: pagerank, GrxPrEngine; device side grx_pr_dist_* in csrc/grx_pr.hip).

CPU part (gloo, world_size 2 and 3): the partition / all-gather / convergence protocol with a numpy engine (test
infrastructure) standing in for the device kernels.  GPU part: the real engine at one rank, and with two ranks sharing
cuda:0 with gloo carrying the exchange (RCCL needs one GPU per rank).  Yardstick: the float64 evaluation of the
reference's recurrence (algorithms/pr.hxx:107-195; oracle orc_pr_f64) for the SAME number of iterations -- tolerance
1e-6 absolute and 1e-4 relative (the reference has no PageRank oracle of its own), and the iteration count
float64 needs (+-1)."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as O
from conftest import ROOT

ALPHA, TOL = 0.85, 1e-6


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class NumpyPrEngine:
    """Host stand-in for GrxPrEngine with the same contract (tests only): numpy float32 versions of the prepare / pack /
    scalar / pull kernels."""
    stream = None

    def __init__(self, out_rows, in_rows, rank, n_ranks):
        import torch
        from gunrock_amd import distributed as D
        self.torch = torch
        self.oro, self.oci, self.ow = out_rows
        self.iro, self.ici, self.iwt = in_rows
        self.rank, self.P = rank, n_ranks
        self.V = len(self.oro) - 1
        self.S = D.slice_bits(self.V, n_ranks)
        self.lo, self.hi = min(rank * self.S, self.V), min((rank + 1) * self.S, self.V)
        self.x = torch.zeros(self.P * self.S, dtype=torch.float32)
        self.pair_out = torch.zeros(2, dtype=torch.int32)
        self.pairs = torch.zeros(2 * self.P, dtype=torch.int32)

    def new_ranks(self):
        return self.torch.zeros(self.S, dtype=self.torch.float32)

    def begin(self, alpha, tol, p_local):
        self.alpha, self.tol = np.float32(alpha), np.float32(tol)
        self.p = p_local.numpy()
        n = self.hi - self.lo
        self.p[:n] = np.float32(1.0 / self.V)
        sums = np.add.reduceat(np.append(self.ow, np.float32(0)).astype(np.float32), self.oro[:-1].astype(np.int64))
        sums[np.diff(self.oro) == 0] = 0
        s = sums[self.lo:self.hi]
        self.iw = np.where(s != 0, self.alpha / np.where(s != 0, s, 1), 0).astype(np.float32)
        self.it, self.done, self.err = 0, False, np.float32(0)

    def pre(self):
        if self.done:
            return
        n = self.hi - self.lo
        self.x.numpy()[self.lo:self.hi] = self.p[:n] * self.iw
        dsum = np.float32((self.alpha * self.p[:n][self.iw == 0]).sum(dtype=np.float32))
        err = self.err if self.it > 0 else np.finfo(np.float32).max
        self.pair_out.numpy()[:] = np.array([dsum, err], dtype=np.float32).view(np.int32)

    def post(self):
        if self.done:
            return
        pairs = self.pairs.numpy().view(np.float32).reshape(self.P, 2)
        if self.it > 0 and pairs[:, 1].max() < self.tol:
            self.done = True
            self.iterations = self.it
            return
        dsum = np.float32(0)
        for r in range(self.P):
            dsum = np.float32(dsum + pairs[r, 0])
        base = np.float32((np.float32(1) - self.alpha + dsum) / np.float32(self.V))
        x = self.x.numpy()
        n = self.hi - self.lo
        new = np.empty(n, np.float32)
        for v in range(self.lo, self.hi):
            a, b = self.iro[v], self.iro[v + 1]
            new[v - self.lo] = base + np.float32((x[self.ici[a:b]] * self.iwt[a:b]).sum(dtype=np.float32))
        self.err = np.float32(np.abs(new - self.p[:n]).max()) if n else np.float32(0)
        self.p[:n] = new
        self.it += 1
        self.iterations = self.it

    def poll(self):
        return self.done, self.it

    def end(self):
        return {"edges_visited": 0, "vertices_visited": 0, "iterations": self.iterations, "elapsed_ms": 0.0}


GRAPHS = (("rmat", 6000, 70000, 5, False), ("rmat_sym", 5000, 40000, 9, True))


def _weights(n, seed):
    return (np.random.default_rng(seed).integers(1, 9, n)).astype(np.float32)


def _worker(rank, world, port, use_gpu, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import gunrock_amd as gr
    from gunrock_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    results = {}
    for kind, V, E, seed, weighted in GRAPHS:
        props, _ = gr.generate(kind, V, E, seed=seed)
        bounds = D.vertex_bounds(V, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        _, mine = gr.generate_rows(kind, V, E, lo, hi, seed=seed)
        _, mine_in = gr.generate_rows(kind, V, E, lo, hi, seed=seed, in_rows=True)
        if weighted:
            # a weight that depends on the edge's endpoints only, so that out-rows and in-rows agree on it
            def wfun(rows_ro, cols):
                src = np.repeat(np.arange(V, dtype=np.int64), np.diff(rows_ro))
                return (1 + (src * 7 + cols.astype(np.int64) * 13) % 5).astype(np.float32), src
            mine.nonzero_values, _ = wfun(mine.row_offsets, mine.column_indices)
            dst = np.repeat(np.arange(V, dtype=np.int64), np.diff(mine_in.row_offsets))
            mine_in.nonzero_values = (1 + (mine_in.column_indices.astype(np.int64) * 7 + dst * 13) % 5).astype(np.float32)
            props.weighted = True
        if use_gpu:
            eng = D.GrxPrEngine(props, mine, mine_in, rank, world, "cuda:0")
        else:
            eng = NumpyPrEngine((mine.row_offsets, mine.column_indices, mine.nonzero_values),
                                (mine_in.row_offsets, mine_in.column_indices, mine_in.nonzero_values), rank, world)
        p, st = D.pagerank(eng, dist, ALPHA, TOL)
        results[kind] = (p.cpu().numpy()[:hi - lo].copy(), lo, hi, st["iterations"])
        del eng
    np.save(os.path.join(out_dir, "pr%d.npy" % rank), np.array([results], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def _full_weighted(gr, kind, V, E, seed, weighted):
    _, full = gr.generate(kind, V, E, seed=seed)
    w = full.nonzero_values
    if weighted:
        src = np.repeat(np.arange(V, dtype=np.int64), np.diff(full.row_offsets))
        w = (1 + (src * 7 + full.column_indices.astype(np.int64) * 13) % 5).astype(np.float32)
    return O.Csr(full.row_offsets, full.column_indices, w)


def _check(gr, per_rank, world):
    for kind, V, E, seed, weighted in GRAPHS:
        g = _full_weighted(gr, kind, V, E, seed, weighted)
        got = np.full(V, np.nan, np.float32)
        its = set()
        for r in range(world):
            part, lo, hi, it = per_rank[r][kind]
            got[lo:hi] = part
            its.add(it)
        assert len(its) == 1, its  # every rank stopped in the same iteration
        it = its.pop()
        _, it64, _ = O.pr_f64(g, ALPHA, TOL)
        assert abs(it - it64) <= 1, (kind, it, it64)
        same_it = O.pr_f64(g, ALPHA, TOL, force_iterations=it)[0]
        assert not np.isnan(got).any()
        assert np.abs(got - same_it).max() <= 1e-6, kind
        nz = same_it > 0
        assert (np.abs(got[nz] - same_it[nz]) / same_it[nz]).max() <= 1e-4, kind


def _run(world, use_gpu, tmp_path):
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(_worker, args=(world, port, use_gpu, str(tmp_path)), nprocs=world, join=True)
    import gunrock_amd as gr
    per_rank = [np.load(os.path.join(str(tmp_path), "pr%d.npy" % r), allow_pickle=True)[0] for r in range(world)]
    _check(gr, per_rank, world)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_pagerank_protocol_on_cpu_with_gloo(world, tmp_path):
    _run(world, False, tmp_path)


@pytest.mark.gpu
def test_pagerank_single_rank_equals_float64_and_plain_engine(gr, gpu_ctx):
    import torch
    from gunrock_amd import distributed as D
    for kind, V, E, seed, weighted in GRAPHS:
        props, full = gr.generate(kind, V, E, seed=seed)
        _, fin = gr.generate_rows(kind, V, E, 0, V, seed=seed, in_rows=True)
        g = _full_weighted(gr, kind, V, E, seed, weighted)
        if weighted:
            full.nonzero_values = g.values
            dst = np.repeat(np.arange(V, dtype=np.int64), np.diff(fin.row_offsets))
            fin.nonzero_values = (1 + (fin.column_indices.astype(np.int64) * 7 + dst * 13) % 5).astype(np.float32)
            props.weighted = True
        eng = D.GrxPrEngine(props, full, fin, 0, 1, "cuda:0")
        p, st = D.pagerank(eng, None, ALPHA, TOL)
        got = p.cpu().numpy()[:V]
        same_it = O.pr_f64(g, ALPHA, TOL, force_iterations=st["iterations"])[0]
        assert np.abs(got - same_it).max() <= 1e-6
        # the single-GPU engine on the same graph: same iteration count, same ranks to 1e-6
        G = gr.build_graph(props, full, gpu_ctx)
        res = gr.pr_result_t(torch.empty(V, dtype=torch.float32, device="cuda:0"))
        gr.pr_run(G, gr.pr_param_t(ALPHA, TOL, None), res, gpu_ctx)
        assert gr.run_stats(gpu_ctx)["search_depth"] == st["iterations"]
        assert np.abs(res.p.cpu().numpy() - got).max() <= 1e-6


@pytest.mark.gpu
def test_pagerank_two_ranks_real_kernels_one_gpu(tmp_path):
    _run(2, True, tmp_path)
