"""Partitioned SSSP (gunrock_amd/distributed.py: sssp, GrxSsspEngine; device side csrc/grx_dist_sssp.hip).

CPU part (gloo, world_size 2 and 3): the partition / all-to-all-of-minima / termination protocol with a numpy engine
(test infrastructure) standing in for the device kernels.  GPU part: the real engine at one rank, and with two ranks
sharing cuda:0 with gloo carrying the exchange (RCCL needs one GPU per rank).  Distances must equal the oracle's
Dijkstra (examples/algorithms/sssp/sssp_cpu.hxx) bit for bit -- the reference's --validate convention."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as O
from conftest import ROOT

FMAX = np.finfo(np.float32).max


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class NumpySsspEngine:
    """Host stand-in for GrxSsspEngine with the same contract (tests only): numpy float32 versions of the head / prep /
    advance / apply / stats kernels."""
    stream = None

    def __init__(self, out_rows, rank, n_ranks):
        import torch
        from gunrock_amd import distributed as D
        self.torch = torch
        self.ro, self.ci, self.w = out_rows
        self.rank, self.P = rank, n_ranks
        self.V = len(self.ro) - 1
        self.S = D.slice_bits(self.V, n_ranks)
        self.lo, self.hi = min(rank * self.S, self.V), min((rank + 1) * self.S, self.V)
        self.send = torch.zeros(self.P * self.S, dtype=torch.float32)
        self.recv = torch.zeros(self.P * self.S, dtype=torch.float32)
        self.stats_local = torch.zeros(4, dtype=torch.int64)
        self.stats_global = torch.zeros(4, dtype=torch.int64)

    def new_labels(self):
        return self.torch.empty(self.S, dtype=self.torch.float32)

    def begin(self, source, distances):
        self.d = distances.numpy()
        self.d[:self.hi - self.lo] = FMAX
        self.level, self.done, self.ev, self.vv = -1, False, 0, 0
        self.frontier = []
        if self.lo <= source < self.hi:
            self.d[source - self.lo] = 0.0
            self.frontier = [source]
        self.stats_local[0] = len(self.frontier)

    def pre(self):
        if self.done:
            return
        if int(self.stats_global[0]) == 0:
            self.done = True
            self.level += 1
            return
        self.level += 1
        send = self.send.numpy()
        send[:] = FMAX
        self.next = set()
        self.vv += len(self.frontier)
        for u in self.frontier:
            du = self.d[u - self.lo]
            for e in range(self.ro[u], self.ro[u + 1]):
                n, nd = int(self.ci[e]), np.float32(du + self.w[e])
                self.ev += 1
                if self.lo <= n < self.hi:
                    if nd < self.d[n - self.lo]:
                        self.d[n - self.lo] = nd
                        self.next.add(n)
                elif nd < send[n]:
                    send[n] = nd

    def post(self):
        if self.done:
            self.stats_local[:] = 0
            return
        recv = self.recv.numpy().reshape(self.P, self.S)
        for j in range(self.P):
            if j == self.rank:
                continue
            better = np.nonzero(recv[j][:self.hi - self.lo] < self.d[:self.hi - self.lo])[0]
            for off in better:
                self.d[off] = recv[j][off]
                self.next.add(self.lo + int(off))
        self.frontier = sorted(self.next)
        self.stats_local[0] = len(self.frontier)

    def poll(self):
        return self.done, self.level

    def end(self):
        return {"edges_visited": self.ev, "vertices_visited": self.vv, "search_depth": self.level, "elapsed_ms": 0.0}


GRAPHS = (("rmat", 4000, 30000, 5), ("rmat_sym", 3000, 16000, 9))


def _weights(row_offsets, cols, V):
    """a weight that depends on the edge's endpoints only: the row slices of every rank agree with the full graph"""
    src = np.repeat(np.arange(V, dtype=np.int64), np.diff(row_offsets))
    return (1 + (src * 7 + cols.astype(np.int64) * 13) % 23).astype(np.float32)


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
    for kind, V, E, seed in GRAPHS:
        props, full = gr.generate(kind, V, E, seed=seed)
        bounds = D.vertex_bounds(V, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        _, mine = gr.generate_rows(kind, V, E, lo, hi, seed=seed)
        for weighted in (True, False):
            if weighted:
                mine.nonzero_values = _weights(mine.row_offsets, mine.column_indices, V)
                mine._device = None
                props.weighted = True
            if use_gpu:
                eng = D.GrxSsspEngine(props, mine, rank, world, "cuda:0")
            else:
                eng = NumpySsspEngine((mine.row_offsets, mine.column_indices, mine.nonzero_values), rank, world)
            d = eng.new_labels()
            for s in (int(np.argmax(np.diff(full.row_offsets))), 0, V - 1):
                st = D.sssp(eng, dist, s, d)
                results["%s_%d_%d" % (kind, int(weighted), s)] = (d.cpu().numpy()[:hi - lo].copy(), lo, hi, st["search_depth"])
            del eng
            _, mine = gr.generate_rows(kind, V, E, lo, hi, seed=seed)
    np.save(os.path.join(out_dir, "ss%d.npy" % rank), np.array([results], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, use_gpu, tmp_path):
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(_worker, args=(world, port, use_gpu, str(tmp_path)), nprocs=world, join=True)
    import gunrock_amd as gr
    per_rank = [np.load(os.path.join(str(tmp_path), "ss%d.npy" % r), allow_pickle=True)[0] for r in range(world)]
    for kind, V, E, seed in GRAPHS:
        _, full = gr.generate(kind, V, E, seed=seed)
        for weighted in (1, 0):
            w = _weights(full.row_offsets, full.column_indices, V) if weighted else full.nonzero_values
            g = O.Csr(full.row_offsets, full.column_indices, w)
            for s in (int(np.argmax(np.diff(full.row_offsets))), 0, V - 1):
                want = O.sssp(g, s)[0]
                got = np.full(V, -1, np.float32)
                depths = set()
                for r in range(world):
                    part, lo, hi, depth = per_rank[r]["%s_%d_%d" % (kind, weighted, s)]
                    got[lo:hi] = part
                    depths.add(depth)
                assert np.array_equal(got, want), (kind, weighted, s)
                assert len(depths) == 1  # every rank stopped in the same iteration


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_sssp_protocol_on_cpu_with_gloo(world, tmp_path):
    _run(world, False, tmp_path)


@pytest.mark.gpu
def test_sssp_single_rank_equals_oracle_and_plain_engine(gr, gpu_ctx):
    import torch
    from gunrock_amd import distributed as D
    for kind, V, E, seed in (("rmat", 60000, 900000, 3), ("rmat_sym", 30000, 200000, 4)):
        props, full = gr.generate(kind, V, E, seed=seed)
        full.nonzero_values = _weights(full.row_offsets, full.column_indices, V)
        props.weighted = True
        g = O.Csr(full.row_offsets, full.column_indices, full.nonzero_values)
        eng = D.GrxSsspEngine(props, full, 0, 1, "cuda:0")
        d = eng.new_labels()
        for s in (int(np.argmax(np.diff(full.row_offsets))), 17):
            st = D.sssp(eng, None, s, d)
            want = O.sssp(g, s)[0]
            assert np.array_equal(d.cpu().numpy()[:V], want)
            G = gr.build_graph(props, full, gpu_ctx)
            dd = torch.empty(V, dtype=torch.float32, device="cuda:0")
            gr.sssp(G, s, dd, None, gpu_ctx)
            assert np.array_equal(dd.cpu().numpy(), want)


@pytest.mark.gpu
def test_sssp_two_ranks_real_kernels_one_gpu(tmp_path):
    _run(2, True, tmp_path)
