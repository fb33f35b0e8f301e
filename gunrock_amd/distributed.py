"""Multi-GPU BFS: one process per GPU, vertex-range partition, RCCL over xGMI
(torch.distributed; backend "nccl" IS RCCL on ROCm).

The reference has no multi-GPU path (every operator throws when
`context.size() != 1`, framework/operators/advance/advance.hxx:129-132); this is
the MI355X design of DESIGN.md section 7 (device side: the second half of csrc/grx_bfs.hip):

  rank r owns the vertex slice [r * S, min((r + 1) * S, V)) -- S = slice_bits(V, P), a
  multiple of 2048 -- with its out-rows and, for the bottom-up step, its in-rows.
  A level group is, per rank and entirely stream-ordered (no host round trip):

    pre   head kernel (termination + Beamer direction from the all-reduced frontier
          statistics, identical on every rank) -> prep -> top-down advance: owned
          targets are claimed locally, targets of other ranks set a bit in the
          outgoing bitmap (one S-bit slice per peer, deduplicated);
          bottom-up level: the rank's frontier slice is replicated to every peer
    a2a   all_to_all_single of the FIXED-SIZE bitmaps (P equal splits of S / 8 bytes;
          each pair of GPUs has its own xGMI link, so the exchange is link-parallel)
    post  top-down: OR of the received slices -> claim -> append to the queue;
          bottom-up: scan in-edges of unvisited owned vertices against the
          whole-graph frontier bitmap the all-to-all assembled; then frontier
          statistics
    ar    all_reduce of 4 int64 words (frontier vertices / out-edges)

  Fixed message sizes mean the host never needs a size: it enqueues level groups
  blindly in batches and polls `done` once per batch, like the single-GPU enactor.
  `overlap=True` cuts a top-down level in two halves with their own bitmaps: the
  all-to-all of the first half is issued asynchronously (RCCL runs it on its own HIP
  stream) while the second half is still advancing on the compute stream.

`engine` abstracts the device side so that the protocol can also be exercised on CPU
tensors with the gloo backend and a numpy engine in tests (tests/test_distributed.py);
the product engine is `GrxEngine` (C ABI grx_bfs_dist_*).
"""
import ctypes as C

import numpy as np

from . import _capi


def slice_bits(n_vertices, n_ranks):
    """Vertices per rank: ceil(V / P) rounded up to a multiple of 2048 (>= 2048)."""
    per = (int(n_vertices) + n_ranks - 1) // n_ranks
    return max(2048, ((per + 2047) // 2048) * 2048)


def vertex_bounds(n_vertices, n_ranks):
    """Slice boundaries of the partition: bounds[r] = min(r * S, V).  (Vertex ids of the
    stand-in graphs are randomly relabelled, so equal vertex counts give statistically
    equal edge counts.)"""
    S = slice_bits(n_vertices, n_ranks)
    return np.asarray([min(r * S, int(n_vertices)) for r in range(n_ranks + 1)], dtype=np.int64).astype(np.int32)


def edge_balanced_bounds(row_offsets, n_ranks):
    """Contiguous vertex ranges with near-equal edge counts (prefix of degrees); a helper
    for callers that place their own data -- the bitmap exchange itself uses vertex_bounds."""
    ro = np.asarray(row_offsets, dtype=np.int64)
    total = int(ro[-1])
    cuts = [0]
    for r in range(1, n_ranks):
        cuts.append(int(np.searchsorted(ro, (total * r) // n_ranks, side="left")))
    cuts.append(len(ro) - 1)
    return np.maximum.accumulate(np.asarray(cuts, dtype=np.int32))


class GrxEngine:
    """Device side: the level-group C ABI (grx_bfs_dist_*).

    The engine context is created ON a torch stream and every torch operation of the
    driver runs under that stream, so kernels, collectives and buffer reads are
    stream-ordered without extra synchronisation."""

    def __init__(self, properties, out_rows, rank, n_ranks, device, n_edges_global, in_rows=None, overlap=False):
        import torch
        from . import build_graph, multi_context_t
        self.torch = torch
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx = multi_context_t(self.device.index or 0, stream=self.stream)
        with torch.cuda.stream(self.stream):
            self.g = build_graph(properties, out_rows, self.ctx, device=device)
            self.g_in = build_graph(properties, in_rows, self.ctx, device=device) if in_rows is not None else None
        self.rank, self.P = int(rank), int(n_ranks)
        self.parts = 2 if overlap else 1
        V = self.g.get_number_of_vertices()
        self.V = V
        self.S = slice_bits(V, self.P)
        self.slice_words = self.S // 32
        words = self.parts * self.P * self.slice_words
        with torch.cuda.stream(self.stream):
            self.send = torch.zeros(words, dtype=torch.int32, device=device)
            self.recv = torch.zeros(words, dtype=torch.int32, device=device)
            self.stats_local = torch.zeros(4, dtype=torch.int64, device=device)
            self.stats_global = torch.zeros(4, dtype=torch.int64, device=device)
        self._h = C.c_void_p()
        _capi.check(_capi.lib().grx_bfs_dist_create(
            self.ctx._h, self.g._h, self.g_in._h if self.g_in is not None else None, self.P, self.rank,
            int(n_edges_global), self.parts, C.c_void_p(self.send.data_ptr()), C.c_void_p(self.recv.data_ptr()),
            C.c_void_p(self.stats_local.data_ptr()), C.c_void_p(self.stats_global.data_ptr()), C.byref(self._h)))

    def new_labels(self):
        """SHARDED label buffer for bfs(): int32[S] on the engine's device, the owned slice only -- vertex v of
        this rank at index v - rank * S (bfs() also accepts a full int32[V] tensor and then writes its owned
        range)."""
        return self.torch.empty(self.S, dtype=self.torch.int32, device=self.device)

    @property
    def lo(self):
        return min(self.rank * self.S, self.V)

    @property
    def hi(self):
        return min((self.rank + 1) * self.S, self.V)

    def disable_library_transport(self):
        """Back to the torch.distributed transport (every rank must take the same decision)."""
        self.library_transport = False

    def group_captured(self):
        """True once the level group of the transport in use has been recorded as a HIP graph."""
        if getattr(self, "library_transport", False):
            try:
                return bool(_capi.lib().grx_bfs_dist_group_is_captured(self._h))
            except Exception:
                return False
        return getattr(self, "_graph", None) is not None

    def transport_description(self):
        if getattr(self, "library_transport", False):
            return ("level groups (kernels + grouped ncclSend/ncclRecv + ncclAllReduce) enqueued by the library itself "
                    "(grx_bfs_dist_groups: one C call per batch, one HIP-graph launch per level after the first search); "
                    "the host polls once per batch of levels")
        return ("level groups (kernels + both collectives) enqueued through torch.distributed and replayed as one "
                "HIP graph per level after the first search; the host polls once per batch of levels")

    def enable_library_transport(self, dist):
        """Let libgrx issue the collectives itself over its own RCCL communicator (include/grx.h, "RCCL transport
        inside the library").  Rank 0 creates the ncclUniqueId, `dist` (torch.distributed, any backend) only
        broadcasts those 128 bytes.  Needs one GPU per rank (RCCL refuses two ranks on one device) and
        overlap=False."""
        torch = self.torch
        L = _capi.lib()
        n = int(L.grx_dist_unique_id_bytes())
        buf = (C.c_ubyte * n)()
        if self.rank == 0:
            _capi.check(L.grx_dist_unique_id(buf))
        if dist is not None and self.P > 1:
            on_gpu = dist.get_backend() == "nccl"
            t = torch.tensor(list(buf), dtype=torch.uint8, device=self.device if on_gpu else "cpu")
            dist.broadcast(t, src=0)
            buf = (C.c_ubyte * n)(*t.cpu().tolist())
        with torch.cuda.stream(self.stream):
            _capi.check(L.grx_bfs_dist_comm_init(self._h, buf))
        self.library_transport = True

    def part_buffers(self, part):
        n = self.P * self.slice_words
        return self.send[part * n:(part + 1) * n], self.recv[part * n:(part + 1) * n]

    def begin(self, source, distances, optimized=True):
        from . import forward, optimized as OPT
        L = _capi.lib()
        n = int(distances.numel())
        if n >= self.V:
            fn = L.grx_bfs_dist_begin        # full-size labels, owned range written in place
        elif n >= self.hi - self.lo:
            fn = L.grx_bfs_dist_begin_local  # sharded labels (new_labels())
        else:
            raise ValueError("distances must hold V labels or at least the owned slice")
        _capi.check(fn(self._h, int(source), int(OPT if optimized else forward), C.c_void_p(distances.data_ptr())))

    def run(self, source, distances, optimized=True):
        """A whole search in ONE C call (grx_bfs_dist_run): with one rank exactly the single-GPU engine; with more, the
        in-library RCCL transport (enable_library_transport) carries the per-level exchange."""
        from . import forward, optimized as OPT
        n = int(distances.numel())
        if n >= self.V:
            local = 0
        elif n >= self.hi - self.lo:
            local = 1
        else:
            raise ValueError("distances must hold V labels or at least the owned slice")
        s = _capi.grx_run_stats_t()
        _capi.check(_capi.lib().grx_bfs_dist_run(self._h, int(source), int(OPT if optimized else forward),
                                                 C.c_void_p(distances.data_ptr()), local, C.byref(s)))
        return {"edges_visited": s.edges_visited, "vertices_visited": s.vertices_visited,
                "search_depth": s.search_depth, "elapsed_ms": s.elapsed_ms}

    def pre(self, part=0):
        _capi.check(_capi.lib().grx_bfs_dist_pre(self._h, int(part)))

    def post(self):
        _capi.check(_capi.lib().grx_bfs_dist_post(self._h))

    def poll(self):
        done, level = C.c_int32(0), C.c_int32(0)
        _capi.check(_capi.lib().grx_bfs_dist_poll(self._h, C.byref(done), C.byref(level)))
        return bool(done.value), level.value

    def end(self):
        s = _capi.grx_run_stats_t()
        _capi.check(_capi.lib().grx_bfs_dist_end(self._h, C.byref(s)))
        return {"edges_visited": s.edges_visited, "vertices_visited": s.vertices_visited,
                "search_depth": s.search_depth, "elapsed_ms": s.elapsed_ms}

    def sync(self):
        self.ctx.synchronize()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _capi.lib().grx_bfs_dist_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _all_to_all(dist, send, recv, async_op=False):
    """Fixed-size bitmap exchange: P equal splits.  nccl (= RCCL): device tensors, runs on
    RCCL's own stream, ordered against the current stream by events.  gloo (tests): through
    host copies."""
    if dist is None or dist.get_world_size() == 1:
        recv.copy_(send)
        return None
    if dist.get_backend() == "nccl":
        return dist.all_to_all_single(recv, send, async_op=async_op)
    s, r = send.cpu(), recv.cpu()
    dist.all_to_all_single(r, s)
    recv.copy_(r)
    return None


def _all_reduce_stats(dist, engine):
    if dist is None or dist.get_world_size() == 1:
        engine.stats_global.copy_(engine.stats_local)
        return
    if dist.get_backend() == "nccl":
        engine.stats_global.copy_(engine.stats_local)
        dist.all_reduce(engine.stats_global)
        return
    t = engine.stats_local.cpu()
    dist.all_reduce(t)
    engine.stats_global.copy_(t)


def _level_group(engine, dist):
    """One level group: pre -> exchange -> post -> statistics all-reduce (all stream-ordered)."""
    works = []
    for part in range(engine.parts):
        engine.pre(part)  # part 0: head + prep + advance of half 0; part 1: advance of half 1
        send, recv = engine.part_buffers(part)
        # asynchronous: the compute stream goes on with the next half while RCCL
        # moves this half's bitmaps on its own stream
        works.append(_all_to_all(dist, send, recv, async_op=engine.parts > 1))
    for w in works:
        if w is not None:
            w.wait()  # stream-level wait, the host does not block
    engine.post()
    _all_reduce_stats(dist, engine)


def profile_bfs(engine, dist, source, distances, optimized=True, max_groups=64):
    """One EAGER search with three stream events per level group -- kernels before the exchange (head, prep,
    advance), the bitmap all-to-all, kernels after it + the statistics all-reduce -- for the N > 1 bench line's per-level
    breakdown.  Untimed diagnostics: every group is followed by a host poll, and the exchange goes through
    torch.distributed whatever transport the timed searches use (same kernels, an equivalent collective).
    -> list of {"pre_ms", "exchange_ms", "post_ms"} per executed level group."""
    torch = engine.torch
    import contextlib
    on_stream = torch.cuda.stream(engine.stream) if getattr(engine, "stream", None) is not None \
        else contextlib.nullcontext()
    recs = []
    with on_stream:
        engine.begin(source, distances, optimized)
        _all_reduce_stats(dist, engine)
        for _ in range(max_groups):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            engine.pre(0)
            ev[1].record()
            send, recv = engine.part_buffers(0)
            _all_to_all(dist, send, recv)
            for part in range(1, engine.parts):
                engine.pre(part)
                s2, r2 = engine.part_buffers(part)
                _all_to_all(dist, s2, r2)
            ev[2].record()
            engine.post()
            _all_reduce_stats(dist, engine)
            ev[3].record()
            done, _ = engine.poll()
            ev[3].synchronize()
            recs.append({"pre_ms": ev[0].elapsed_time(ev[1]), "exchange_ms": ev[1].elapsed_time(ev[2]),
                         "post_ms": ev[2].elapsed_time(ev[3])})
            if done:
                break
        engine.end()
    return recs


def _group_graph(engine, dist, key):
    """A level group takes no level-dependent argument, so it can be captured ONCE into a HIP graph
    -- five kernel launches and the two collectives (RCCL collectives are capturable) -- and every
    later group is one graph launch: the host side of a level drops from seven Python / ctypes /
    torch.distributed calls (~100 us) to one (~15 us), which is what bounds the search once the
    exchange sits between the kernels.  Captured while no search is running (capture records, it does
    not execute); the same on every rank.  Any failure falls back to eager calls for good."""
    import os
    torch = engine.torch
    if getattr(engine, "_graph_failed", False) or os.environ.get("GRX_DIST_GRAPH", "1") == "0":
        return None
    if getattr(engine, "stream", None) is None or engine.parts != 1:
        return None
    if dist is not None and dist.get_world_size() > 1 and dist.get_backend() != "nccl":
        return None
    cached = getattr(engine, "_graph", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=engine.stream, capture_error_mode="thread_local"):
            _level_group(engine, dist)
        engine._graph = (key, g)
        return g
    except Exception:  # capture not possible here: eager from now on
        engine._graph_failed = True
        engine._graph = None
        return None


def _first_batch(engine, default):
    """Groups enqueued before the first look at `done`.  Every rank must enqueue the SAME number of groups (a group
    carries two collectives), so the schedule may only depend on values all ranks share: here the depth of the previous
    search on this engine (global by construction).  A repeated or similar search then costs exactly its groups and one
    poll -- the doubling schedule from 4 spent 12 groups and two polls on a 7-level search."""
    last = getattr(engine, "_last_groups", 0)
    return max(2, min(int(last), 64)) if last else default


def _remember_groups(engine, stats):
    engine._last_groups = int(stats["search_depth"]) + 1  # + the group whose head finds the frontier empty
    return stats


def bfs(engine, dist, source, distances, optimized=True, first_batch=4):
    """Partitioned BFS driven by this rank.  `distances`: int32 tensor on the engine's device, either
    SHARDED (engine.new_labels(): the owned slice only, vertex v at v - engine.lo) or full-size (V entries,
    the owned range is written); on return it holds the depths of the owned vertices.  `dist`: the
    torch.distributed module (or None for a single rank).  Returns run stats (edges /
    vertices are this rank's share; search_depth is global)."""
    torch = engine.torch
    import contextlib
    on_stream = torch.cuda.stream(engine.stream) if getattr(engine, "stream", None) is not None \
        else contextlib.nullcontext()
    import os
    whole = hasattr(engine, "run") and os.environ.get("GRX_DIST_LEVEL_GROUPS", "0") != "1"
    if whole and (engine.P == 1 or getattr(engine, "library_transport", False)):
        # ONE C call per search (grx_bfs_dist_run).  One rank: the partition of one slice is the graph, and the search is the
        # single-GPU engine's, launch schedule included.  More ranks: the library runs the collectives itself -- begin, the
        # seed's all-reduce, batches of level groups with one look at `done` each, end.
        with on_stream:
            return _remember_groups(engine, engine.run(source, distances, optimized))
    if getattr(engine, "library_transport", False):
        # (GRX_DIST_LEVEL_GROUPS=1: the same from Python, one C call per batch of level groups)
        L = _capi.lib()
        with on_stream:
            engine.begin(source, distances, optimized)
            _capi.check(L.grx_bfs_dist_seed_stats(engine._h))
            batch = _first_batch(engine, first_batch)
            while True:
                _capi.check(L.grx_bfs_dist_groups(engine._h, batch))
                done, _ = engine.poll()
                if done:
                    break
                batch = min(batch * 2, 32)
            _capi.check(L.grx_bfs_dist_capture_group(engine._h))  # no-op once recorded for this buffer / direction
            return _remember_groups(engine, engine.end())
    with on_stream:
        engine.begin(source, distances, optimized)
        _all_reduce_stats(dist, engine)
        key = (int(distances.data_ptr()), bool(optimized))
        cached = getattr(engine, "_graph", None)
        graph = cached[1] if (cached is not None and cached[0] == key) else None
        batch = _first_batch(engine, first_batch)
        while True:
            for _ in range(batch):
                if graph is not None:
                    graph.replay()
                else:
                    _level_group(engine, dist)
            done, _ = engine.poll()
            if done:
                break
            batch = min(batch * 2, 32)
        if graph is None:
            # the search is over (every kernel of a further group would exit on `done`): record the
            # group for the next search with the same label buffer and direction setting
            _group_graph(engine, dist, key)
        return _remember_groups(engine, engine.end())


# ------------------------------------------------------------------------------------------------------------------
# Partitioned PageRank (device side: grx_pr_dist_* in csrc/grx_pr.hip).
#   rank r owns the vertices [r * S, min((r + 1) * S, V)) with their out-rows (inverse weight sums) and in-rows (the
#   pull), p is sharded.  One iteration, per rank and stream-ordered:
#     pre   x[v] = p[v] * iweights[v] for the owned v, into this rank's slice of the global x buffer (vertex v sits at
#           x[v]: slices are S apart and S-aligned); the rank's {dangling sum, norm of the previous iteration}
#     ag    all_gather of the x slices (S floats per rank) and of the pairs (2 words per rank)
#     post  convergence test + base term from the gathered pairs, summed / maximised in rank order on every rank (so
#           all ranks stop in the same iteration without talking to the host), then the pull over the owned rows
#   The host enqueues iterations blindly in batches and polls `done` once per batch; iterations after `done` are no-ops.


class GrxPrEngine:
    """Device side of the partitioned PageRank: the C ABI grx_pr_dist_*."""

    def __init__(self, properties, out_rows, in_rows, rank, n_ranks, device):
        import torch
        from . import build_graph, multi_context_t
        self.torch = torch
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx = multi_context_t(self.device.index or 0, stream=self.stream)
        with torch.cuda.stream(self.stream):
            self.g = build_graph(properties, out_rows, self.ctx, device=device)
            self.g_in = build_graph(properties, in_rows, self.ctx, device=device)
        self.rank, self.P = int(rank), int(n_ranks)
        self.V = self.g.get_number_of_vertices()
        self.S = slice_bits(self.V, self.P)
        self.lo, self.hi = min(self.rank * self.S, self.V), min((self.rank + 1) * self.S, self.V)
        with torch.cuda.stream(self.stream):
            self.x = torch.zeros(self.P * self.S, dtype=torch.float32, device=device)
            self.pair_out = torch.zeros(2, dtype=torch.int32, device=device)
            self.pairs = torch.zeros(2 * self.P, dtype=torch.int32, device=device)
        self._h = C.c_void_p()
        _capi.check(_capi.lib().grx_pr_dist_create(self.ctx._h, self.g._h, self.g_in._h, self.P, self.rank,
                                                   self.lo, self.hi, C.byref(self._h)))

    def new_ranks(self):
        """SHARDED result buffer: float32[S], vertex v of this rank at index v - rank * S"""
        return self.torch.empty(self.S, dtype=self.torch.float32, device=self.device)

    def begin(self, alpha, tol, p_local):
        assert p_local.dtype == self.torch.float32 and p_local.numel() >= self.hi - self.lo
        self._p = p_local
        _capi.check(_capi.lib().grx_pr_dist_begin(
            self._h, float(alpha), float(tol), C.c_void_p(p_local.data_ptr()), C.c_void_p(self.x.data_ptr()),
            C.c_void_p(self.pair_out.data_ptr()), C.c_void_p(self.pairs.data_ptr())))

    def pre(self):
        _capi.check(_capi.lib().grx_pr_dist_pre(self._h))

    def post(self):
        _capi.check(_capi.lib().grx_pr_dist_post(self._h))

    def poll(self):
        done, it = C.c_int32(0), C.c_int32(0)
        _capi.check(_capi.lib().grx_pr_dist_poll(self._h, C.byref(done), C.byref(it)))
        return bool(done.value), it.value

    def end(self):
        s = _capi.grx_run_stats_t()
        _capi.check(_capi.lib().grx_pr_dist_end(self._h, C.byref(s)))
        return {"edges_visited": s.edges_visited, "vertices_visited": s.vertices_visited,
                "iterations": s.search_depth, "elapsed_ms": s.elapsed_ms}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _capi.lib().grx_pr_dist_destroy(h)
            self._h = None


def _all_gather_slices(dist, buf, mine_off, n):
    """every rank's [r * n, (r + 1) * n) of `buf` <- that rank's own slice (this rank's slice is already in place)"""
    if dist is None or dist.get_world_size() == 1:
        return
    mine = buf[mine_off:mine_off + n]
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(buf, mine)  # in place: the input IS the rank's slice of the output
    else:
        outs = [buf[r * n:(r + 1) * n] for r in range(dist.get_world_size())]
        dist.all_gather(outs, mine.clone())


def pagerank(engine, dist, alpha=0.85, tol=1e-6, p_local=None, first_batch=4, max_iterations=1000):
    """Partitioned PageRank driven by this rank.  `p_local`: float32 tensor on the engine's device holding the OWNED slice
    (engine.new_ranks()); on return it holds the ranks of the owned vertices.  Returns (p_local, stats); stats["iterations"]
    is global (every rank stops in the same iteration)."""
    torch = engine.torch
    import contextlib
    on_stream = torch.cuda.stream(engine.stream) if getattr(engine, "stream", None) is not None \
        else contextlib.nullcontext()
    if p_local is None:
        p_local = engine.new_ranks()
    with on_stream:
        engine.begin(alpha, tol, p_local)
        launched, batch = 0, first_batch
        while True:
            for _ in range(batch):
                engine.pre()
                _all_gather_slices(dist, engine.x, engine.rank * engine.S, engine.S)
                engine.pairs[2 * engine.rank:2 * engine.rank + 2].copy_(engine.pair_out)
                _all_gather_slices(dist, engine.pairs, 2 * engine.rank, 2)
                engine.post()
                launched += 1
            done, _ = engine.poll()
            if done or launched >= max_iterations:
                break
            batch = min(batch * 2, 16)
        return p_local, engine.end()


# ------------------------------------------------------------------------------------------------------------------
# Partitioned SSSP (device side: csrc/grx_dist_sssp.hip).  The level-group pattern of the BFS with a float
# payload: per iteration every pair of ranks exchanges S floats -- the smallest tentative distance the sender found
# for each of the receiver's vertices (FLT_MAX: none) -- and one all-reduce carries the size of the next frontier.


class GrxSsspEngine:
    """Device side of the partitioned SSSP: the C ABI grx_sssp_dist_*."""

    def __init__(self, properties, out_rows, rank, n_ranks, device):
        import torch
        from . import build_graph, multi_context_t
        self.torch = torch
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx = multi_context_t(self.device.index or 0, stream=self.stream)
        with torch.cuda.stream(self.stream):
            self.g = build_graph(properties, out_rows, self.ctx, device=device)
        self.rank, self.P = int(rank), int(n_ranks)
        self.V = self.g.get_number_of_vertices()
        self.S = slice_bits(self.V, self.P)
        self.lo, self.hi = min(self.rank * self.S, self.V), min((self.rank + 1) * self.S, self.V)
        with torch.cuda.stream(self.stream):
            self.send = torch.zeros(self.P * self.S, dtype=torch.float32, device=device)
            self.recv = torch.zeros(self.P * self.S, dtype=torch.float32, device=device)
            self.stats_local = torch.zeros(4, dtype=torch.int64, device=device)
            self.stats_global = torch.zeros(4, dtype=torch.int64, device=device)
        self._h = C.c_void_p()
        _capi.check(_capi.lib().grx_sssp_dist_create(
            self.ctx._h, self.g._h, self.P, self.rank, C.c_void_p(self.send.data_ptr()), C.c_void_p(self.recv.data_ptr()),
            C.c_void_p(self.stats_local.data_ptr()), C.c_void_p(self.stats_global.data_ptr()), C.byref(self._h)))

    def new_labels(self):
        """SHARDED distance buffer: float32[S], vertex v of this rank at index v - rank * S"""
        return self.torch.empty(self.S, dtype=self.torch.float32, device=self.device)

    def begin(self, source, distances):
        assert distances.dtype == self.torch.float32 and distances.numel() >= self.hi - self.lo
        self._d = distances
        _capi.check(_capi.lib().grx_sssp_dist_begin(self._h, int(source), C.c_void_p(distances.data_ptr())))

    def pre(self):
        _capi.check(_capi.lib().grx_sssp_dist_pre(self._h))

    def post(self):
        _capi.check(_capi.lib().grx_sssp_dist_post(self._h))

    def poll(self):
        done, level = C.c_int32(0), C.c_int32(0)
        _capi.check(_capi.lib().grx_sssp_dist_poll(self._h, C.byref(done), C.byref(level)))
        return bool(done.value), level.value

    def end(self):
        s = _capi.grx_run_stats_t()
        _capi.check(_capi.lib().grx_sssp_dist_end(self._h, C.byref(s)))
        return {"edges_visited": s.edges_visited, "vertices_visited": s.vertices_visited,
                "search_depth": s.search_depth, "elapsed_ms": s.elapsed_ms}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _capi.lib().grx_sssp_dist_destroy(h)
            self._h = None


def sssp(engine, dist, source, distances, first_batch=4):
    """Partitioned SSSP driven by this rank.  `distances`: float32 tensor on the engine's device holding the OWNED slice
    (engine.new_labels()); on return it holds the distances of the owned vertices (FLT_MAX: unreachable).  Returns run
    stats (edges / vertices are this rank's share; search_depth = iterations, global)."""
    torch = engine.torch
    import contextlib
    on_stream = torch.cuda.stream(engine.stream) if getattr(engine, "stream", None) is not None \
        else contextlib.nullcontext()
    with on_stream:
        engine.begin(source, distances)
        _all_reduce_stats(dist, engine)
        batch = _first_batch(engine, first_batch)
        while True:
            for _ in range(batch):
                engine.pre()
                _all_to_all(dist, engine.send, engine.recv)
                engine.post()
                _all_reduce_stats(dist, engine)
            done, _ = engine.poll()
            if done:
                break
            batch = min(batch * 2, 32)
        return _remember_groups(engine, engine.end())
