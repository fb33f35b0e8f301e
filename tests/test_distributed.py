"""Multi-GPU BFS (gunrock_amd/distributed.py).

CPU part (gloo, world_size 2 and 3): the partition / all-to-all / termination
protocol with a FAKE engine (numpy top-down step, test infrastructure) standing
in for the device kernels.  GPU part: the real engine (grx_bfs_dist_* kernels)
with two ranks sharing cuda:0 and gloo carrying the exchange -- RCCL itself
needs one GPU per rank, which only the driver's multi-GPU box has."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as O
from conftest import ROOT

INF = np.iinfo(np.int32).max


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeEngine:
    """Host stand-in for GrxEngine with the same contract (tests only)."""

    def __init__(self, ro, ci, bounds, rank):
        import torch
        self.torch = torch
        self.ro, self.ci, self.bounds, self.rank = ro, ci, np.asarray(bounds), rank
        self.P = len(bounds) - 1
        self.V = len(ro) - 1
        self.send = torch.zeros(self.V, dtype=torch.int32)
        self.counts = torch.zeros(self.P, dtype=torch.int64)

    def begin(self, source, dist):
        self.dist = dist.numpy()
        self.dist[:] = INF
        lo, hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.frontier_v = []
        self.level = -1
        self.ev = 0
        if lo <= source < hi:
            self.dist[source] = 0
            self.frontier_v = [int(source)]

    def advance(self):
        self.level += 1
        lo, hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        nxt, remote = [], [[] for _ in range(self.P)]
        for u in self.frontier_v:
            self.ev += self.ro[u + 1] - self.ro[u]
            for e in range(self.ro[u], self.ro[u + 1]):
                n = int(self.ci[e])
                if self.dist[n] > self.level + 1:
                    self.dist[n] = self.level + 1
                    if lo <= n < hi:
                        nxt.append(n)
                    else:
                        remote[int(np.searchsorted(self.bounds, n, side="right") - 1)].append(n)
        self.next_v = nxt
        send = self.send.numpy()
        for j in range(self.P):
            self.counts[j] = len(remote[j])
            send[self.bounds[j]: self.bounds[j] + len(remote[j])] = remote[j]
        return self.send, self.counts

    def apply(self, recv, n):
        for v in recv[:n].tolist():
            if self.dist[v] > self.level + 1:
                self.dist[v] = self.level + 1
                self.next_v.append(v)

    def frontier(self):
        self.frontier_v = self.next_v
        return len(self.frontier_v), 0

    def end(self):
        return {"edges_visited": int(self.ev), "vertices_visited": 0, "search_depth": self.level + 1, "elapsed_ms": 0.0}


def _worker(rank, world, port, use_gpu, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import gunrock_amd as gr
    from gunrock_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V, E = 20000, 160000
    results = {}
    for kind, seed in (("rmat", 5), ("rmat_sym", 9)):
        props, full = gr.generate(kind, V, E, seed=seed)
        bounds = D.vertex_bounds(V, world) if kind == "rmat" else D.edge_balanced_bounds(full.row_offsets, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        _, mine = gr.generate_rows(kind, V, E, lo, hi, seed=seed)
        src = int(np.argmax(np.diff(full.row_offsets)))
        if use_gpu:
            eng = D.GrxEngine(props, mine, bounds, rank, "cuda:0")
            d = torch.empty(V, dtype=torch.int32, device="cuda:0")
        else:
            eng = FakeEngine(mine.row_offsets, mine.column_indices, bounds, rank)
            d = torch.empty(V, dtype=torch.int32)
        for s in (src, 0, V - 1):
            st = D.bfs(eng, dist, s, d, bounds, rank)
            results["%s_%d" % (kind, s)] = (d.cpu().numpy()[lo:hi].copy(), lo, hi, st["edges_visited"], st["search_depth"])
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([results], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, use_gpu, tmp_path):
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(_worker, args=(world, port, use_gpu, str(tmp_path)), nprocs=world, join=True)
    import gunrock_amd as gr
    per_rank = [np.load(os.path.join(str(tmp_path), "r%d.npy" % r), allow_pickle=True)[0] for r in range(world)]
    V, E = 20000, 160000
    for kind, seed in (("rmat", 5), ("rmat_sym", 9)):
        _, full = gr.generate(kind, V, E, seed=seed)
        g = O.Csr(full.row_offsets, full.column_indices, full.nonzero_values)
        src = int(np.argmax(np.diff(full.row_offsets)))
        for s in (src, 0, V - 1):
            want, _, ev = O.bfs_queue(g, s)
            got = np.full(V, -1, np.int32)
            edges = 0
            for r in range(world):
                part, lo, hi, e_r, depth = per_rank[r]["%s_%d" % (kind, s)]
                got[lo:hi] = part
                edges += e_r
            assert np.array_equal(got, want), (kind, s)
            assert edges == ev  # every reached vertex is expanded exactly once, by its owner
            assert depth == want[want != INF].max() + 1


@pytest.mark.parametrize("world", [2, 3])
def test_protocol_on_cpu_with_gloo(world, tmp_path):
    _run(world, False, tmp_path)


def test_partition_helpers(gr):
    from gunrock_amd import distributed as D
    b = D.vertex_bounds(10, 3)
    assert b.tolist() == [0, 3, 6, 10]
    _, c = gr.generate("rmat", 5000, 60000, seed=1)
    e = D.edge_balanced_bounds(c.row_offsets, 4)
    assert e[0] == 0 and e[-1] == 5000 and np.all(np.diff(e) >= 0)
    per = [c.row_offsets[e[i + 1]] - c.row_offsets[e[i]] for i in range(4)]
    assert max(per) < 1.35 * (60000 / 4)
    # row slices of the generator tile the full graph exactly
    _, a = gr.generate_rows("rmat", 5000, 60000, 0, int(e[2]), seed=1)
    _, b2 = gr.generate_rows("rmat", 5000, 60000, int(e[2]), 5000, seed=1)
    assert np.array_equal(np.concatenate([a.column_indices, b2.column_indices]), c.column_indices)


@pytest.mark.gpu
def test_two_ranks_real_kernels_one_gpu(tmp_path):
    _run(2, True, tmp_path)


@pytest.mark.gpu
def test_single_rank_dist_path_equals_plain_bfs(gr, gpu_ctx):
    import torch
    from gunrock_amd import distributed as D

    class NoDist:
        @staticmethod
        def get_backend():
            return "none"
    V, E = 1 << 16, 1 << 20
    props, c = gr.generate("rmat", V, E, seed=2)
    bounds = D.vertex_bounds(V, 1)
    eng = D.GrxEngine(props, c, bounds, 0, "cuda:0")
    d = torch.empty(V, dtype=torch.int32, device="cuda:0")
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = int(np.argmax(np.diff(g.row_offsets)))
    st = D.bfs(eng, NoDist, src, d, bounds, 0)
    want, _, ev = O.bfs_queue(g, src)
    assert np.array_equal(d.cpu().numpy(), want) and st["edges_visited"] == ev


@pytest.mark.gpu
def test_bench_multi_rank_path(tmp_path):
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run), except
    that both ranks share cuda:0 and gloo carries the exchange."""
    import json
    import subprocess
    env = dict(os.environ, GRX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "small"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["n_vertices"] == 2 * (1 << 18)
