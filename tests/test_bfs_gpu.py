"""BFS parity on the GPU, through the C ABI (gunrock_amd -> libgrx.so):
bit-exact depths against the oracle and the reference-produced goldens."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
INF = np.iinfo(np.int32).max


def run_bfs(gr, ctx, ro, ci, src, options=None, w=None):
    import torch
    csr = gr.csr_t.from_arrays(ro, ci, w)
    G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=False, symmetric=False), csr, ctx)
    dist = torch.full((G.get_number_of_vertices(),), -7, dtype=torch.int32, device="cuda:0")
    pred = torch.empty_like(dist)
    ms = gr.bfs(G, src, dist, pred, ctx, options)
    assert ms >= 0
    return dist.cpu().numpy(), gr.run_stats(ctx)


def all_options(gr):
    yield None
    for lb in (gr.thread_mapped, gr.warp_mapped, gr.block_mapped, gr.merge_path, gr.merge_path_v2, gr.bucketing):
        yield gr.options_t(advance_load_balance=lb)
    yield gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact)
    yield gr.options_t(engine_flags=gr.FLAG_SYNC_EACH_LEVEL)
    yield gr.options_t(advance_direction=gr.optimized)  # direction-optimising (bottom-up fat levels)
    yield gr.options_t(advance_direction=gr.optimized, engine_flags=gr.FLAG_PROFILE)
    yield gr.options_t(engine_flags=1 << 8)  # tuning variant: bitmap claim
    yield gr.options_t(engine_flags=gr.FLAG_PROFILE)


def test_chesapeake_golden(gr, gpu_ctx, golden):
    props, coo = gr.matrix_market_t().load(os.path.join(GOLDEN, "chesapeake.mtx"))
    csr = gr.csr_t().from_coo(coo)
    for opt in all_options(gr):
        for s in (0, 5, 38):
            d, st = run_bfs(gr, gpu_ctx, csr.row_offsets, csr.column_indices, s, opt)
            assert np.array_equal(d, golden["chesapeake_bfs_%d" % s])
            assert st["search_depth"] == d.max() + 1


def test_synthetic_goldens(gr, gpu_ctx, golden):
    for opt in all_options(gr):
        d, st = run_bfs(gr, gpu_ctx, golden["rmat_ro"], golden["rmat_ci"], int(golden["rmat_src"][0]), opt)
        assert np.array_equal(d, golden["rmat_bfs"])
        d, _ = run_bfs(gr, gpu_ctx, golden["road_ro"], golden["road_ci"], int(golden["road_src"][0]), opt)
        assert np.array_equal(d, golden["road_bfs"])


def test_edges_visited_matches_definition(gr, gpu_ctx, golden):
    # edges_visited = sum of out-degrees of reached vertices (benchmark.hxx LOG_EDGE_VISITED semantics)
    g = O.Csr(golden["rmat_ro"], golden["rmat_ci"], np.ones(len(golden["rmat_ci"]), np.float32))
    src = int(golden["rmat_src"][0])
    d, st = run_bfs(gr, gpu_ctx, g.row_offsets, g.column_indices, src)
    _, _, ev = O.bfs_queue(g, src)
    assert st["edges_visited"] == ev
    assert st["vertices_visited"] == int((d != INF).sum())


def test_edge_cases(gr, gpu_ctx):
    # single vertex, no edges
    d, st = run_bfs(gr, gpu_ctx, [0, 0], [], 0)
    assert d.tolist() == [0]
    # isolated source in a bigger graph; self loops and duplicate edges
    ro = np.array([0, 0, 3, 5, 5, 6], dtype=np.int32)
    ci = np.array([1, 2, 2, 3, 3, 1], dtype=np.int32)
    g = O.Csr(ro, ci, np.ones(6, np.float32))
    for s in range(5):
        d, _ = run_bfs(gr, gpu_ctx, ro, ci, s)
        assert np.array_equal(d, O.bfs(g, s)[0])
    # source out of range -> error, like an exception in the reference
    with pytest.raises(gr.GrxError):
        run_bfs(gr, gpu_ctx, ro, ci, 5)
    with pytest.raises(gr.GrxError):
        run_bfs(gr, gpu_ctx, ro, ci, 0, gr.options_t(advance_load_balance=gr.work_stealing))


def test_star_and_chain(gr, gpu_ctx):
    # one hub with 100k leaves (many chunks from one tile), leaves link back
    n = 100_001
    ro = np.concatenate([[0, n - 1], n - 1 + np.arange(1, n)]).astype(np.int32)
    ci = np.concatenate([np.arange(1, n), np.zeros(n - 1)]).astype(np.int32)
    g = O.Csr(ro, ci, np.ones(len(ci), np.float32))
    for s in (0, 77):
        d, _ = run_bfs(gr, gpu_ctx, ro, ci, s)
        assert np.array_equal(d, O.bfs_queue(g, s)[0])
    # long chain: 3000 levels of frontier size 1
    n = 3000
    ro = np.minimum(np.arange(n + 1), n - 1).astype(np.int32)
    ci = np.arange(1, n).astype(np.int32)
    d, st = run_bfs(gr, gpu_ctx, ro, ci, 0)
    assert np.array_equal(d, np.arange(n))
    assert st["search_depth"] == n


def test_random_graphs_vs_oracle(gr, gpu_ctx):
    rng = np.random.default_rng(11)
    for trial in range(6):
        V = int(rng.integers(300, 20000))
        E = int(rng.integers(V, 12 * V))
        _, c = gr.generate("rmat", V, E, seed=100 + trial)
        g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
        for src in (int(np.argmax(np.diff(g.row_offsets))), int(rng.integers(0, V))):
            d, _ = run_bfs(gr, gpu_ctx, g.row_offsets, g.column_indices, src)
            assert np.array_equal(d, O.bfs(g, src)[0])


def test_direction_optimizing_switches_and_matches(gr, gpu_ctx):
    """Symmetric (in-edges = CSR) and directed (engine builds the transpose) graphs;
    the fat levels must actually run bottom-up and depths / counters must not change."""
    import torch
    for kind, V, E in (("rmat_sym", 1 << 17, 1_500_000), ("rmat", 1 << 17, 3_000_000)):
        props, c = gr.generate(kind, V, E, seed=77)
        g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
        src = int(np.argmax(np.diff(g.row_offsets)))
        want, _, ev = O.bfs_queue(g, src)
        G = gr.build_graph(props, c, gpu_ctx)
        dist = torch.empty(V, dtype=torch.int32, device="cuda:0")
        gr.bfs(G, src, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.optimized, engine_flags=gr.FLAG_PROFILE))
        st = gr.run_stats(gpu_ctx)
        prof = gr.level_profile(gpu_ctx)
        assert np.array_equal(dist.cpu().numpy(), want)
        assert st["edges_visited"] == ev and st["vertices_visited"] == int((want != INF).sum())
        assert any(l["bottom_up"] for l in prof), "no level ran bottom-up"
        assert all(l["bu_probes"] > 0 and l["bu_open"] > 0 for l in prof if l["bottom_up"])
        assert prof[0]["frontier_size"] == 1  # the first level is always top-down
        for s2 in (0, 17, V - 1):  # low-degree / isolated sources: may never switch
            gr.bfs(G, s2, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.optimized))
            assert np.array_equal(dist.cpu().numpy(), O.bfs_queue(g, s2)[0])


def test_medium_rmat_and_repeatability(gr, gpu_ctx):
    _, c = gr.generate("rmat", 1 << 18, 4_000_000, seed=42)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = int(np.argmax(np.diff(g.row_offsets)))
    want, _, ev = O.bfs_queue(g, src)
    for i, opt in enumerate(all_options(gr)):
        d, st = run_bfs(gr, gpu_ctx, g.row_offsets, g.column_indices, src, opt)
        assert np.array_equal(d, want), i
        assert st["edges_visited"] == ev, (i, st, ev)


def test_full_size_livejournal_standin_properties(gr, gpu_ctx):
    """BASELINE.json configs[1] size (4,847,571 V / 68,993,773 E): the oracle's
    exact fixed-point characterisation (orc_check_bfs) instead of a second CPU run."""
    _, c = gr.generate("rmat", 4_847_571, 68_993_773, seed=42)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = int(np.argmax(np.diff(g.row_offsets)))
    for opt in (None, gr.options_t(advance_direction=gr.optimized)):
        d, st = run_bfs(gr, gpu_ctx, g.row_offsets, g.column_indices, src, opt)
        assert O.check_bfs(g, src, d) == 0
        reached = d != INF
        assert st["vertices_visited"] == int(reached.sum())
        assert st["edges_visited"] == int(np.diff(g.row_offsets)[reached].sum())


@pytest.mark.gpu
def test_full_size_twitter_standin_properties(gr, gpu_ctx):
    """BASELINE.json configs[4] size on ONE GPU (C5': 21,297,772 V / ~530 M E after symmetric doubling): forward
    (binned fat levels) and direction-optimising searches against the oracle's exact fixed-point characterisation,
    counters against the degrees of the reached set, and the two directions against each other."""
    _, c = gr.generate("rmat_sym", 21_297_772, 265_025_809, seed=42)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    assert g.n_edges > 520_000_000
    src = int(np.argmax(np.diff(g.row_offsets)))
    got = []
    for opt in (gr.options_t(advance_direction=gr.forward), gr.options_t(advance_direction=gr.optimized)):
        d, st = run_bfs(gr, gpu_ctx, g.row_offsets, g.column_indices, src, opt)
        assert O.check_bfs(g, src, d) == 0
        reached = d != INF
        assert st["vertices_visited"] == int(reached.sum())
        assert st["edges_visited"] == int(np.diff(g.row_offsets)[reached].sum())
        got.append(d)
    assert np.array_equal(got[0], got[1])


@pytest.mark.gpu
def test_regression_frontier_merge_persisted_in_bottomup(gr, gpu_ctx):
    """Found by tests/tools/fuzz_gpu.py: a vertex discovered top-down sits in the frontier bitmap
    only; the bottom-up level that consumes that frontier must write the merged word back to
    `visited` even when its 64-vertex chunk discovers nothing (here: the last, 2-vertex chunk),
    or the next bottom-up level re-discovers the vertex at a greater depth."""
    import torch
    props, c = gr.generate("rmat_sym", 317250, 6588144, seed=716928211)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    G = gr.build_graph(props, c, gpu_ctx)
    V = G.get_number_of_vertices()
    dist = torch.empty(V, dtype=torch.int32, device="cuda")
    for src in (133258, 194002, int(np.argmax(np.diff(g.row_offsets)))):
        if src >= V:
            continue
        want, _ = O.bfs(g, src)
        for _ in range(2):
            gr.bfs(G, src, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.optimized))
            assert np.array_equal(dist.cpu().numpy(), want), src


def test_binned_forward_levels(gr, gpu_ctx, monkeypatch, golden):
    """Forward-only BFS with its fat levels run as the binned scatter + claim kernels (grx_bin.hpp):
    GRX_BIN_MIN_EDGES=1 forces EVERY level through them, the default threshold mixes them with the
    claim-per-edge advance and the LDS-resident tiny levels; depths and counters must not change.
    Covers bins of one bitmap word (V = 39), ragged last bins, hubs split over many chunks, duplicate
    edges and self loops, and the A/B knobs (no pre-filter bitmap, no bins, no bitmap at all)."""
    import torch
    cases = []
    props, coo = gr.matrix_market_t().load(os.path.join(GOLDEN, "chesapeake.mtx"))
    cs = gr.csr_t().from_coo(coo)
    cases.append((O.Csr(cs.row_offsets, cs.column_indices, cs.nonzero_values), [0, 5, 38]))
    for kind, V, E, seed in (("rmat", 1 << 17, 3_000_000, 5), ("rmat_sym", 100_003, 1_200_000, 6),
                             ("rmat", 8191, 200_000, 7), ("rmat_sym", 1 << 18, 6_000_000, 8)):
        _, c = gr.generate(kind, V, E, seed=seed)
        g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
        cases.append((g, [int(np.argmax(np.diff(g.row_offsets))), 1, g.n_vertices - 1]))
    for g, sources in cases:
        assert g.n_edges >= 4 * g.n_vertices  # dense enough for the forward bitmap path
        for src in sources:
            want, _, ev = O.bfs_queue(g, src)
            for env in ({"GRX_BIN_MIN_EDGES": "1", "GRX_BIN_MAX_DEGREE": "0"}, {},
                        {"GRX_BIN_MIN_EDGES": "1", "GRX_BIN_MAX_DEGREE": "0", "GRX_TD_PRE": "1"},
                        {"GRX_BIN_MIN_EDGES": "1000"}, {"GRX_TD_BIN": "0"}, {"GRX_TD_BITMAP": "0"}):
                for k in ("GRX_BIN_MIN_EDGES", "GRX_BIN_MAX_DEGREE", "GRX_TD_PRE", "GRX_TD_BIN", "GRX_TD_BITMAP"):
                    monkeypatch.delenv(k, raising=False)
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                for flags in (0, gr.FLAG_PROFILE):
                    d, st = run_bfs(gr, gpu_ctx, g.row_offsets, g.column_indices, src,
                                    gr.options_t(advance_direction=gr.forward, engine_flags=flags))
                    assert np.array_equal(d, want), (g.n_vertices, src, env, flags)
                    assert st["edges_visited"] == ev and st["vertices_visited"] == int((want != INF).sum())
                    if flags and env.get("GRX_BIN_MAX_DEGREE") == "0" and "GRX_TD_BIN" not in env:
                        prof = gr.level_profile(gpu_ctx)
                        assert all(l["bottom_up"] == 2 for l in prof if l["edges"] > 0), "a level did not run binned"


def test_binned_kernels_follow_the_previous_search(gr, gpu_ctx, monkeypatch):
    """The scatter / sweep kernels of the binned levels are launched only in the groups where the PREVIOUS forward search
    on the graph handle met a fat level (one group of slack): searches from other sources, whose fat levels fall elsewhere,
    must still give the oracle's depths (such a level runs on the claim-per-edge advance) and re-train the hint."""
    import torch
    _, c = gr.generate("rmat_sym", 1 << 18, 6_000_000, seed=11)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=False, symmetric=False),
                       gr.csr_t.from_arrays(g.row_offsets, g.column_indices, None), gpu_ctx)
    deg = np.diff(g.row_offsets)
    hub = int(np.argmax(deg))
    # a chain hung onto the graph would be nicer; low-degree sources shift the fat levels by one or two groups
    lows = [int(v) for v in np.nonzero(deg == 1)[0][:2]] + [int(np.nonzero(deg > 0)[0][-1])]
    monkeypatch.setenv("GRX_BIN_MIN_EDGES", "100000")
    dist = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
    binned = {}
    for src in [hub, hub, lows[0], lows[0], hub, lows[1], lows[2], hub, hub]:
        want, _, ev = O.bfs_queue(g, src)
        for flags in (0, gr.FLAG_PROFILE):
            gr.bfs(G, src, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.forward, engine_flags=flags))
            assert np.array_equal(dist.cpu().numpy(), want), (src, flags)
            assert gr.run_stats(gpu_ctx)["edges_visited"] == ev
        binned.setdefault(src, []).append(sum(1 for l in gr.level_profile(gpu_ctx) if l["bottom_up"] == 2))
    # the same source twice in a row: the second search bins every level the first one wanted binned
    assert binned[hub][1] >= 1 and binned[hub][-1] == binned[hub][1], binned
    monkeypatch.setenv("GRX_BIN_HINT", "0")
    gr.bfs(G, hub, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.forward, engine_flags=gr.FLAG_PROFILE))
    assert sum(1 for l in gr.level_profile(gpu_ctx) if l["bottom_up"] == 2) >= binned[hub][1]


def test_exact_schedule_of_a_repeated_source_and_uniform_bins(gr, gpu_ctx, monkeypatch):
    """Round 5.  (a) A forward search from the SAME source as the last one on the handle launches exactly the kernels that search
    needed (grx_graph::bin_exact: fat groups without a level kernel, thin groups without scatter + sweep); a prediction that no
    longer holds -- the binning threshold moved between two searches -- must only be slow: a thin level in a group without a
    level kernel is binned, a fat one in a group without the two kernels runs on the claim-per-edge advance.  (b) Both bin cuts
    -- the balanced one with its granule table and the uniform 2^k-vertex ranges (bin = id >> k) -- give the oracle's depths."""
    import torch
    _, c = gr.generate("rmat_sym", 1 << 19, 9_000_000, seed=5)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    deg = np.diff(g.row_offsets)
    hub = int(np.argmax(deg))
    other = int(np.nonzero(deg == 2)[0][0])
    want = {s: O.bfs_queue(g, s) for s in (hub, other)}
    dist = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
    for uniform in ("0", "1"):
        monkeypatch.setenv("GRX_BIN_UNIFORM", uniform)
        monkeypatch.setenv("GRX_BIN_MIN_EDGES", "200000")
        G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=False, symmetric=False),
                           gr.csr_t.from_arrays(g.row_offsets, g.column_indices, None), gpu_ctx)
        launches = []
        for src, flags, min_edges in ((hub, 0, 200000), (hub, gr.FLAG_ASYNC_RETURN, 200000), (hub, 0, 200000),
                                      (hub, 0, 1 << 30),          # nothing is fat any more: the fat groups bin thin levels
                                      (hub, 0, 1 << 30), (hub, 0, 20000),  # more levels fat than predicted
                                      (hub, 0, 20000), (other, 0, 20000), (hub, 0, 20000), (hub, gr.FLAG_PROFILE, 20000), (hub, 0, 20000)):
            monkeypatch.setenv("GRX_BIN_MIN_EDGES", str(min_edges))
            gr.bfs(G, src, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.forward, engine_flags=flags))
            gpu_ctx.synchronize()
            assert np.array_equal(dist.cpu().numpy(), want[src][0]), (uniform, src, flags, min_edges)
            st = gr.run_stats(gpu_ctx)
            assert st["edges_visited"] == want[src][2]
            launches.append(int(st["aux"]))
        # the group in which the previous search from this source ended is launched as its head alone; a search that does not
        # end in that group after all (here: the prediction is forced to be too short) must continue in the next one
        for forced in ("1", "2", "3", "4"):
            monkeypatch.setenv("GRX_GROUP_HINT_FORCE", forced)
            for flags in (0, gr.FLAG_ASYNC_RETURN):
                gr.bfs(G, hub, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.forward, engine_flags=flags))
                gpu_ctx.synchronize()
                assert np.array_equal(dist.cpu().numpy(), want[hub][0]), (uniform, forced, flags)
        monkeypatch.delenv("GRX_GROUP_HINT_FORCE")
        # the same for a repeated direction-optimising search (grx_graph::do_last_src; its head finishes or leaves the search alone)
        for forced in (None, None, "1", "2", "3", "4", None):
            if forced is None:
                monkeypatch.delenv("GRX_GROUP_HINT_FORCE", raising=False)
            else:
                monkeypatch.setenv("GRX_GROUP_HINT_FORCE", forced)
            for flags in (0, gr.FLAG_ASYNC_RETURN):
                gr.bfs(G, hub, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.optimized, engine_flags=flags))
                gpu_ctx.synchronize()
                assert np.array_equal(dist.cpu().numpy(), want[hub][0]), (uniform, "optimized", forced, flags)
        monkeypatch.delenv("GRX_GROUP_HINT_FORCE", raising=False)
        monkeypatch.setenv("GRX_BIN_EXACT", "0")
        gr.bfs(G, hub, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.forward))
        assert np.array_equal(dist.cpu().numpy(), want[hub][0])
        monkeypatch.delenv("GRX_BIN_EXACT")
        del G


def test_symmetric_property_is_verified(gr, gpu_ctx, monkeypatch):
    """graph_properties_t defaults to symmetric=true (inert in the reference).  Round 4: the bottom-up step always reads the
    transpose (stable sort, hubs first); with GRX_BU_SYMMETRIC_CSR=1 a graph declared symmetric uses its own CSR as the
    in-edge list (rounds 1-3) -- so the engine must then notice a DIRECTED CSR that claims to be symmetric (row hashes of
    the CSR against the transpose) and keep the transpose."""
    import torch
    _, c = gr.generate("rmat", 1 << 16, 1_500_000, seed=21)  # directed
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = int(np.argmax(np.diff(g.row_offsets)))
    want, _ = O.bfs(g, src)
    _, cs = gr.generate("rmat_sym", 1 << 16, 700_000, seed=22)  # really symmetric
    gs = O.Csr(cs.row_offsets, cs.column_indices, cs.nonzero_values)
    src_s = int(np.argmax(np.diff(gs.row_offsets)))
    want_s, _ = O.bfs(gs, src_s)
    n = 4096  # a directed cycle, whose in- and out-degrees all agree
    ro = np.arange(n + 1, dtype=np.int32)
    ci = ((np.arange(n) + 1) % n).astype(np.int32)
    for env in ("0", "1"):
        monkeypatch.setenv("GRX_BU_SYMMETRIC_CSR", env)
        G = gr.build_graph(gr.graph_properties_t(), c, gpu_ctx)  # default properties: symmetric = True
        dist = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
        gr.bfs(G, src, dist, None, gpu_ctx, gr.options_t(advance_direction=gr.optimized, engine_flags=gr.FLAG_PROFILE))
        assert np.array_equal(dist.cpu().numpy(), want), env
        assert any(l["bottom_up"] == 1 for l in gr.level_profile(gpu_ctx))
        Gs = gr.build_graph(gr.graph_properties_t(), cs, gpu_ctx)
        ds = torch.empty(gs.n_vertices, dtype=torch.int32, device="cuda:0")
        gr.bfs(Gs, src_s, ds, None, gpu_ctx, gr.options_t(advance_direction=gr.optimized))
        assert np.array_equal(ds.cpu().numpy(), want_s), env
        d, _ = run_bfs(gr, gpu_ctx, ro, ci, 0, gr.options_t(advance_direction=gr.optimized))
        assert np.array_equal(d, np.arange(n)), env


def test_binned_scatter_unit_hand_out_safety_net(gr, monkeypatch):
    """The second scatter draws its work units from per-XCD ticket queues, which is only complete when every XCD runs a
    workgroup of the launch.  (a) GRX_SC2_STATIC=1: statically strided units give the same depths.
    (b) GRX_SC2_FAULT_XCD=1 makes the workgroups of one XCD take no units: the sweep's coverage check (bins must hold
    exactly the level's out-edges) stops the search, the host repeats it with static units -- the caller sees correct
    depths, and the context stays in static mode (the fault knob is ignored from then on)."""
    import torch
    _, c = gr.generate("rmat", 1 << 18, 6_000_000, seed=13)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = int(np.argmax(np.diff(g.row_offsets)))
    want, _, ev = O.bfs_queue(g, src)
    monkeypatch.setenv("GRX_BIN_MIN_EDGES", "100000")
    o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact)
    for env in ({"GRX_SC2_STATIC": "1"}, {"GRX_SC2_FAULT_XCD": "1"}, {"GRX_SC2_FAULT_XCD": "3"}):
        ctx = gr.multi_context_t(0)  # a fresh context: the static mode is sticky per context
        G = gr.build_graph(gr.graph_properties_t(True, False, False), gr.csr_t.from_arrays(g.row_offsets, g.column_indices), ctx)
        d = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
        for k in ("GRX_SC2_STATIC", "GRX_SC2_FAULT_XCD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for rep in range(3):
            for flags in (0, gr.FLAG_ASYNC_RETURN):
                o.engine_flags = flags
                gr.bfs(G, src, d, None, ctx, o)
                ctx.synchronize()
                assert np.array_equal(d.cpu().numpy(), want), (env, rep, flags)
                assert gr.run_stats(ctx)["edges_visited"] == ev
        del G, ctx


def test_two_contexts_search_one_graph_handle_concurrently(gr, gpu_ctx):
    """Per-search scratch (bin candidate array, fill / ticket words, bitmaps, queues) belongs to the CONTEXT; a graph handle
    only carries immutable per-graph tables.  Two host threads, two contexts (two streams), one handle: forward (binned fat
    levels) and direction-optimising searches from different sources at the same time must both give the oracle's depths
    (the reference's batch operator runs run() from N host threads, operators/batch/batch.hxx:70-75)."""
    import threading
    import torch
    _, c = gr.generate("rmat_sym", 1 << 18, 5_000_000, seed=17)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    G = gr.build_graph(gr.graph_properties_t(True, False, True), gr.csr_t.from_arrays(g.row_offsets, g.column_indices), gpu_ctx)
    deg = np.diff(g.row_offsets)
    srcs = [int(v) for v in np.argsort(deg)[-4:]]
    want = {s: O.bfs_queue(g, s)[0] for s in srcs}
    warm = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
    for direction in (gr.forward, gr.optimized):  # per-graph preprocessing once, before the threads start
        gr.bfs(G, srcs[0], warm, None, gpu_ctx, gr.options_t(advance_direction=direction))
    errors = []

    def worker(idx):
        try:
            ctx = gr.multi_context_t(0)
            d = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
            for rep in range(12):
                s = srcs[(idx * 2 + rep) % len(srcs)]
                direction = gr.forward if (rep + idx) % 3 else gr.optimized
                gr.bfs(G, s, d, None, ctx, gr.options_t(advance_load_balance=gr.merge_path, advance_direction=direction))
                if not np.array_equal(d.cpu().numpy(), want[s]):
                    errors.append((idx, rep, s, int(direction)))
        except Exception as e:  # noqa: BLE001
            errors.append((idx, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_first_searches_of_two_contexts_race_on_a_fresh_handle(gr, gpu_ctx):
    """Round 5: the lazy per-graph builds -- transpose, bin table, two-neighbour array, weight statistics, pull
    layout -- are serialised by the handle's build lock and published only when complete.  Two contexts make their FIRST
    searches (direction-optimising BFS, forward BFS, SSSP, PageRank) on a handle nobody has prepared, at the same time."""
    import threading
    import torch
    _, c = gr.generate("rmat_sym", 1 << 17, 3_000_000, seed=41)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = int(np.argmax(np.diff(g.row_offsets)))
    want, _ = O.bfs(g, src)
    for trial in range(3):
        G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=False, symmetric=False),
                           gr.csr_t.from_arrays(g.row_offsets, g.column_indices, None), gpu_ctx)
        errors = []
        barrier = threading.Barrier(2)

        def worker(idx):
            try:
                ctx = gr.multi_context_t(0)
                d = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
                f = torch.empty(g.n_vertices, dtype=torch.float32, device="cuda:0")
                barrier.wait()
                order = (gr.optimized, gr.forward) if (idx + trial) % 2 else (gr.forward, gr.optimized)
                for direction in order:
                    gr.bfs(G, src, d, None, ctx, gr.options_t(advance_direction=direction))
                    if not np.array_equal(d.cpu().numpy(), want):
                        errors.append((idx, "bfs", int(direction)))
                gr.sssp(G, src, f, None, ctx, gr.options_t())
                reached = want != np.iinfo(np.int32).max
                if not np.array_equal(f.cpu().numpy()[reached], want[reached].astype(np.float32)):
                    errors.append((idx, "sssp"))
                gr.pr_run(G, gr.pr_param_t(0.85, 1e-6), gr.pr_result_t(f), ctx)
            except Exception as e:  # noqa: BLE001
                errors.append((idx, repr(e)))

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        del G


def test_direction_optimising_search_without_a_transpose_runs_forward(gr, gpu_ctx, monkeypatch):
    """Round 5: when the transpose cannot be built (its sort needs 16-24 transient bytes per edge) a
    direction-optimising search runs forward-only instead of failing; the next call, with memory to spare, builds it."""
    import torch
    _, c = gr.generate("rmat_sym", 1 << 16, 1_500_000, seed=43)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = int(np.argmax(np.diff(g.row_offsets)))
    want, _ = O.bfs(g, src)
    G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=False, symmetric=True),
                       gr.csr_t.from_arrays(g.row_offsets, g.column_indices, None), gpu_ctx)
    d = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
    monkeypatch.setenv("GRX_TR_FAIL_ALLOC", "1")
    for flags in (0, gr.FLAG_PROFILE):
        gr.bfs(G, src, d, None, gpu_ctx, gr.options_t(advance_direction=gr.optimized, engine_flags=flags))
        assert np.array_equal(d.cpu().numpy(), want)
    assert all(l["bottom_up"] != 1 for l in gr.level_profile(gpu_ctx))  # no bottom-up level ran
    monkeypatch.delenv("GRX_TR_FAIL_ALLOC")
    gr.bfs(G, src, d, None, gpu_ctx, gr.options_t(advance_direction=gr.optimized, engine_flags=gr.FLAG_PROFILE))
    assert np.array_equal(d.cpu().numpy(), want)
    assert any(l["bottom_up"] == 1 for l in gr.level_profile(gpu_ctx))


def _with_tail(g, anchor, length):
    """the graph plus a path of `length` new vertices hung onto `anchor` (edges both ways): a source at its end is `length`
    levels deeper than anything else"""
    V = g.n_vertices
    src_e = np.repeat(np.arange(V, dtype=np.int64), np.diff(g.row_offsets))
    dst_e = g.column_indices.astype(np.int64)
    a = np.concatenate([[anchor], V + np.arange(length - 1)])
    b = V + np.arange(length)
    s2 = np.concatenate([src_e, a, b])
    d2 = np.concatenate([dst_e, b, a])
    order = np.argsort(s2, kind="stable")
    ro = np.zeros(V + length + 1, np.int64)
    np.cumsum(np.bincount(s2, minlength=V + length), out=ro[1:])
    ci = d2[order].astype(np.int32)
    return O.Csr(ro.astype(np.int32), ci, np.ones(len(ci), np.float32))


def test_launch_groups_follow_the_previous_search(gr, gpu_ctx, monkeypatch):
    """Paced searches enqueue as many launch groups as the previous search on the graph handle needed and then WAIT for the
    end (run_levels: hold_after) -- a search that is longer than the prediction (the stream drains without `done`), shorter,
    or equal must give the oracle's depths, blocking and with GRX_FLAG_ASYNC_RETURN, forward and direction-optimising; so
    must the one-launch reset + seed of a forward search and the source kernel that writes level 1's chunk map
    (GRX_FWD_SEED_IN_RESET, GRX_SOURCE_MAP: on / off)."""
    import torch
    _, c = gr.generate("rmat_sym", 1 << 18, 6_000_000, seed=21)
    g0 = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    deg = np.diff(g0.row_offsets)
    hub = int(np.argmax(deg))
    anchor = int(np.nonzero(deg == 1)[0][0])
    g = _with_tail(g0, anchor, 9)
    far = g.n_vertices - 1
    assert g.n_edges >= 8 * g.n_vertices  # paced enqueueing
    G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=False, symmetric=True),
                       gr.csr_t.from_arrays(g.row_offsets, g.column_indices, None), gpu_ctx)
    monkeypatch.setenv("GRX_BIN_MIN_EDGES", "100000")
    want = {s: O.bfs_queue(g, s) for s in (hub, far, anchor)}
    dist = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
    groups = {}  # launch groups of the LAST of the repeated hub searches, per (hint on?, direction)
    for env in ({}, {"GRX_GROUP_HINT": "0"}, {"GRX_FWD_SEED_IN_RESET": "0"}, {"GRX_SOURCE_MAP": "0"},
                {"GRX_FWD_SEED_IN_RESET": "0", "GRX_SOURCE_MAP": "0", "GRX_GROUP_HINT": "0"}):
        for k in ("GRX_GROUP_HINT", "GRX_FWD_SEED_IN_RESET", "GRX_SOURCE_MAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for direction in (gr.forward, gr.optimized):
            for flags in (0, gr.FLAG_ASYNC_RETURN):
                # short, long (prediction too short), long (exact), short (prediction too long), a third depth, short
                for src in (hub, far, far, hub, anchor, hub, hub):
                    dist.fill_(-5)
                    gr.bfs(G, src, dist, None, gpu_ctx, gr.options_t(advance_direction=direction, engine_flags=flags))
                    gpu_ctx.synchronize()
                    got = dist.cpu().numpy()
                    assert np.array_equal(got, want[src][0]), (env, direction, flags, src)
                    if not flags:
                        st = gr.run_stats(gpu_ctx)
                        assert st["edges_visited"] == want[src][2]
                        if len(env) <= 1 and "GRX_FWD_SEED_IN_RESET" not in env and "GRX_SOURCE_MAP" not in env:
                            groups[(env.get("GRX_GROUP_HINT", "1"), direction)] = int(st["aux"])
    # a repeated search launches exactly the groups it needs with the prediction, and never fewer without it
    for direction in (gr.forward, gr.optimized):
        assert 1 <= groups[("1", direction)] <= groups[("0", direction)], groups


def test_plan_rules_for_searches_from_low_degree_sources(gr, gpu_ctx, monkeypatch):
    """The two plan rules of round 5's last session (grx_frontier.hpp plan_in::mid_tile_e / bin_early_div): a frontier
    with a tile of more than 4096 out-edges does not enter the many-levels body (two hops from a low-degree source of a
    scale-free graph: a dozen vertices with 20-45 k out-edges, which ONE workgroup of that body would walk alone), and a level
    met while less than a quarter of the graph is visited is binned from half the usual number of out-edges.  Either rule only
    chooses between bodies that are all exact: depths and counters equal the oracle's with the rules on and off; for the
    second rule the profile shows which body ran."""
    _, c = gr.generate("rmat", 1 << 19, 8_000_000, seed=21)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    deg = np.diff(g.row_offsets).astype(np.int64)
    knobs = ("GRX_MID_TILE_E", "GRX_BIN_MIN_EDGES", "GRX_BIN_EARLY_DIV", "GRX_MID_HUB_DEG")
    # sources 8 and 15: level 2 is 15 / 11 vertices with 44920 / 20588 out-edges; source 3: level 1 is 3 vertices with 4443;
    # source 0: level 2 is 399 vertices with 491807 out-edges, 401 vertices visited by then
    for src, heavy_level, early_level in ((8, 2, None), (15, 2, None), (3, 1, None), (0, None, 2)):
        want, _, ev = O.bfs_queue(g, src)
        reached = want != INF
        ne = np.bincount(want[reached], weights=deg[reached]).astype(np.int64)
        nv = np.bincount(want[reached])
        if heavy_level is not None:
            assert nv[heavy_level] <= 256 and 4096 < ne[heavy_level] <= 65536, "the generator changed: pick another source"
        if early_level is not None:
            assert 400_000 <= ne[early_level] < 800_000 and nv[:early_level + 1].sum() * 4 < g.n_vertices
        for env in ({}, {"GRX_MID_TILE_E": "0"}, {"GRX_MID_HUB_DEG": "0"}, {"GRX_BIN_MIN_EDGES": "800000"},
                    {"GRX_BIN_MIN_EDGES": "800000", "GRX_BIN_EARLY_DIV": "1"}):
            for k in knobs:
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for flags in (0, gr.FLAG_PROFILE):
                d, st = run_bfs(gr, gpu_ctx, g.row_offsets, g.column_indices, src,
                                gr.options_t(advance_direction=gr.forward, engine_flags=flags))
                assert np.array_equal(d, want), (src, env, flags)
                assert st["edges_visited"] == ev and st["vertices_visited"] == int(reached.sum()), (src, env, flags)
            # Which body ran is visible in the profile only for the second rule: a PROFILED run has no tiny levels in its heads, so
            # level 0 itself enters the many-levels body and that body keeps going level after level on its own exit rules (total
            # vertices / out-edges) -- the first rule is a rule of the heads (profiles/r5_c36_ms_trace.txt has the kernel sequences
            # of unprofiled searches with and without it).
            prof = gr.level_profile(gpu_ctx)  # of the profiled run
            modes = {int(l["edges"]): int(l["bottom_up"]) for l in prof}
            if early_level is not None and env.get("GRX_BIN_MIN_EDGES") == "800000":
                m = modes.get(int(ne[early_level]))
                assert m == (0 if env.get("GRX_BIN_EARLY_DIV") == "1" else 2), (src, env, prof[:5])
            # Round 6: the body's own EXIT rule by mean out-degree (pipe_args::mid_hub_deg, GRX_MID_HUB_DEG; the patch round 5
            # left unapplied).  A profiled search enters the body at level 0; with the rule it LEAVES in front of the level of
            # hubs, which then is a record of its own on the regular level kernel (mode 0); without it (GRX_MID_HUB_DEG=0) the
            # body walks that level itself -- no record with that level's edge count, or one of mode 3.
            if heavy_level is not None and "GRX_BIN_MIN_EDGES" not in env and ne[heavy_level] > 16 * nv[heavy_level]:
                m = modes.get(int(ne[heavy_level]))
                if env.get("GRX_MID_HUB_DEG") == "0":
                    assert m is None or m == 3, (src, env, prof[:5])
                elif "GRX_MID_TILE_E" not in env:
                    assert m == 0, (src, env, prof[:5])
    for k in knobs:
        monkeypatch.delenv(k, raising=False)
