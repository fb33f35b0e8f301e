"""Multi-GPU BFS: one process per GPU, vertex-range partition, RCCL all-to-all
frontier exchange (torch.distributed; backend "nccl" is RCCL on ROCm).

The reference has no multi-GPU path (every operator throws when
`context.size() != 1`, framework/operators/advance/advance.hxx:129-132); this is
the MI355X design of DESIGN.md section 6:

  rank r owns vertices [bounds[r], bounds[r+1]) and their CSR rows (global column
  ids).  Per level:
    1. local advance (the single-GPU fused kernel) over the owned frontier;
    2. winners not owned by this rank are binned by owner on the device;
    3. bucket sizes, then buckets, are exchanged with ALL-TO-ALL (each pair of GPUs
       uses its own xGMI link, so the exchange is link-parallel);
    4. received candidates are claimed (atomicMin) and appended to the next frontier;
    5. the global next-frontier size is all-reduced for termination.
  Depths are authoritative on the owned range of every rank.

`engine` abstracts the device side (grx_bfs_dist_* in include/grx.h) so that the
protocol can also be exercised on CPU tensors with the gloo backend and a fake
engine in tests (tests/test_distributed.py); the product engine is `GrxEngine`.
"""
import ctypes as C

import numpy as np

from . import _capi


def vertex_bounds(n_vertices, n_ranks):
    """Contiguous, near-equal vertex ranges (vertex ids of the stand-in graphs are
    randomly relabelled, so equal vertex counts give statistically equal edge counts)."""
    b = [(n_vertices * r) // n_ranks for r in range(n_ranks + 1)]
    return np.asarray(b, dtype=np.int32)


def edge_balanced_bounds(row_offsets, n_ranks):
    """Contiguous vertex ranges with near-equal edge counts (prefix of degrees)."""
    ro = np.asarray(row_offsets, dtype=np.int64)
    total = int(ro[-1])
    cuts = [0]
    for r in range(1, n_ranks):
        cuts.append(int(np.searchsorted(ro, (total * r) // n_ranks, side="left")))
    cuts.append(len(ro) - 1)
    return np.maximum.accumulate(np.asarray(cuts, dtype=np.int32))


class GrxEngine:
    """Device side: the level-stepping C ABI (grx_bfs_dist_*).

    The engine context is created ON a torch stream, and every torch operation of the
    exchange runs on that same stream, so kernels, bucket reads and collectives are
    stream-ordered without extra synchronisation."""

    def __init__(self, properties, csr_rows, bounds, rank, device):
        import torch
        from . import build_graph, multi_context_t
        self.torch = torch
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx = multi_context_t(self.device.index or 0, stream=self.stream)
        with torch.cuda.stream(self.stream):
            self.g = build_graph(properties, csr_rows, self.ctx, device=device)
        self.rank = rank
        self.bounds = np.ascontiguousarray(bounds, dtype=np.int32)
        self.P = len(self.bounds) - 1
        V = self.g.get_number_of_vertices()
        self.send = torch.empty(max(V, 1), dtype=torch.int32, device=device)
        self.counts = torch.zeros(self.P, dtype=torch.int64, device=device)

    def begin(self, source, dist):
        lo, hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        src = int(source) if lo <= source < hi else -1
        _capi.check(_capi.lib().grx_bfs_dist_begin(
            self.ctx._h, self.g._h, src, self.bounds.ctypes.data_as(C.POINTER(C.c_int32)), self.P, self.rank,
            C.c_void_p(self.send.data_ptr()), C.c_void_p(dist.data_ptr())))

    def advance(self):
        """-> (send buffer, device int64[P] bucket sizes); bucket j at send[bounds[j]:]"""
        _capi.check(_capi.lib().grx_bfs_dist_advance(self.ctx._h, C.c_void_p(self.counts.data_ptr())))
        return self.send, self.counts

    def apply(self, recv, n):
        if n > 0:
            _capi.check(_capi.lib().grx_bfs_dist_apply(self.ctx._h, C.c_void_p(recv.data_ptr()), int(n)))

    def frontier(self):
        nv, ne = C.c_longlong(0), C.c_longlong(0)
        _capi.check(_capi.lib().grx_bfs_dist_frontier(self.ctx._h, C.byref(nv), C.byref(ne)))
        return nv.value, ne.value

    def end(self):
        s = _capi.grx_run_stats_t()
        _capi.check(_capi.lib().grx_bfs_dist_end(self.ctx._h, C.byref(s)))
        return {"edges_visited": s.edges_visited, "vertices_visited": s.vertices_visited,
                "search_depth": s.search_depth, "elapsed_ms": s.elapsed_ms}

    def sync(self):
        self.ctx.synchronize()


def _exchange(dist, engine, send, counts, bounds, rank, P, recv):
    """All-to-all of the per-owner buckets.  Returns the number of received ids,
    packed at the front of `recv`."""
    torch = engine.torch
    counts_host = counts.to("cpu")  # the one host read of this level (bucket sizes)
    backend = dist.get_backend()
    if backend == "nccl":
        got = torch.empty_like(counts)
        dist.all_to_all_single(got, counts)
        got_host = got.to("cpu").tolist()
        sent = counts_host.tolist()
        ins = [send[int(bounds[j]): int(bounds[j]) + int(sent[j])] for j in range(P)]
        outs, at = [], 0
        for j in range(P):
            outs.append(recv[at: at + int(got_host[j])])
            at += int(got_host[j])
        dist.all_to_all(outs, ins)  # RCCL: one send/recv pair per peer, each on its own xGMI link
        return at
    # gloo (CPU tensors, tests): sizes by all_gather, payload by point-to-point
    table = [torch.empty(P, dtype=torch.int64) for _ in range(P)]
    dist.all_gather(table, counts_host)
    sent = counts_host.tolist()
    send_cpu = send.to("cpu")
    reqs, pieces = [], []
    at = 0
    for j in range(P):
        n_in = int(table[j][rank])
        piece = torch.empty(n_in, dtype=torch.int32)
        pieces.append(piece)
        if j == rank:
            continue
        if n_in:
            reqs.append(dist.irecv(piece, src=j))
        n_out = int(sent[j])
        if n_out:
            reqs.append(dist.isend(send_cpu[int(bounds[j]): int(bounds[j]) + n_out].contiguous(), dst=j))
    for r in reqs:
        r.wait()
    for j in range(P):
        if j != rank and len(pieces[j]):
            recv[at: at + len(pieces[j])] = pieces[j].to(recv.device)
            at += len(pieces[j])
    return at


def bfs(engine, dist, source, distances, bounds, rank, recv=None):
    """Partitioned BFS driven by this rank.  `distances`: full-size int32 tensor on the
    engine's device; on return its owned range holds the depths.  Returns run stats
    (edges/vertices are this rank's share; search_depth is global)."""
    torch = engine.torch
    P = len(bounds) - 1
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    if recv is None:
        recv = torch.empty(max(hi - lo, 1) * max(P - 1, 1), dtype=torch.int32, device=distances.device)
    import contextlib
    on_stream = torch.cuda.stream(engine.stream) if getattr(engine, "stream", None) is not None \
        else contextlib.nullcontext()
    with on_stream:
        engine.begin(source, distances)
        levels = 0
        while True:
            send, counts = engine.advance()
            n_recv = _exchange(dist, engine, send, counts, bounds, rank, P, recv) if P > 1 else 0
            engine.apply(recv, n_recv)
            nv, _ = engine.frontier()
            total = torch.tensor([nv], dtype=torch.int64,
                                 device=distances.device if dist.get_backend() == "nccl" else "cpu")
            if P > 1:
                dist.all_reduce(total)
            levels += 1
            if int(total.item()) == 0:
                break
        stats = engine.end()
    stats["search_depth"] = levels
    return stats
