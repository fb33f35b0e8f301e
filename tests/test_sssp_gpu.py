"""SSSP parity on the GPU through the C ABI: distances bit-exact (fp32 ==)
against the oracle / reference CPU Dijkstra -- the reference's own --validate
convention (exact != on floats, util/compare.hxx:21)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max


def run_sssp(gr, ctx, ro, ci, w, src, options=None, weighted=True):
    import torch
    csr = gr.csr_t.from_arrays(ro, ci, w)
    G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=weighted, symmetric=False), csr, ctx)
    dist = torch.full((G.get_number_of_vertices(),), -1.0, dtype=torch.float32, device="cuda:0")
    pred = torch.empty(G.get_number_of_vertices(), dtype=torch.int32, device="cuda:0")
    gr.sssp(G, src, dist, pred, ctx, options)
    return dist.cpu().numpy(), gr.run_stats(ctx)


def test_goldens(gr, gpu_ctx, golden):
    for s in (0, 5, 38):
        d, _ = run_sssp(gr, gpu_ctx, golden["chesapeake_ro"], golden["chesapeake_ci"], golden["chesapeake_w"], s)
        assert np.array_equal(d, golden["chesapeake_sssp_%d" % s])
    d, _ = run_sssp(gr, gpu_ctx, golden["tiny_ro"], golden["tiny_ci"], golden["tiny_w"], 0)
    assert np.array_equal(d, golden["tiny_sssp_0"])
    d, _ = run_sssp(gr, gpu_ctx, golden["tsym_ro"], golden["tsym_ci"], golden["tsym_w"], 0)
    assert np.array_equal(d, golden["tsym_sssp_0"])
    d, _ = run_sssp(gr, gpu_ctx, golden["road_ro"], golden["road_ci"], golden["road_w"], int(golden["road_src"][0]))
    assert np.array_equal(d, golden["road_sssp"])


def test_pattern_graph_equals_bfs_depths(gr, gpu_ctx, golden):
    # pattern => unit weights => SSSP == BFS depths as floats; values=None path
    ro, ci = golden["rmat_ro"], golden["rmat_ci"]
    src = int(golden["rmat_src"][0])
    import torch
    csr = gr.csr_t.from_arrays(ro, ci)
    G = gr.graph_t(gr.graph_properties_t(True, False, False),
                   (csr.to_device()[0], csr.to_device()[1], None), gpu_ctx)
    dist = torch.empty(len(ro) - 1, dtype=torch.float32, device="cuda:0")
    gr.sssp(G, src, dist, None, gpu_ctx)
    want = golden["rmat_bfs"].astype(np.float64)
    want[golden["rmat_bfs"] == np.iinfo(np.int32).max] = FMAX
    assert np.array_equal(dist.cpu().numpy(), want.astype(np.float32))


def test_random_weighted_graphs_vs_oracle(gr, gpu_ctx):
    rng = np.random.default_rng(5)
    for trial in range(6):
        V = int(rng.integers(200, 30000))
        E = int(rng.integers(V, 10 * V))
        _, c = gr.generate("rmat", V, E, seed=500 + trial)
        # non-trivial fp32 weights (not exactly representable sums)
        w = (rng.random(E, dtype=np.float32) * np.float32(9.7) + np.float32(0.01)).astype(np.float32)
        g = O.Csr(c.row_offsets, c.column_indices, w)
        for src in (int(np.argmax(np.diff(g.row_offsets))), int(rng.integers(0, V))):
            for opt in (None, gr.options_t(advance_load_balance=gr.merge_path),
                        gr.options_t(advance_load_balance=gr.warp_mapped, enable_uniquify=True),
                        gr.options_t(engine_flags=0x10),   # 0x10: plain label-correcting schedule
                        gr.options_t(engine_flags=0x20)):  # 0x20: near-far even on dense graphs
                d, _ = run_sssp(gr, gpu_ctx, g.row_offsets, g.column_indices, w, src, opt)
                assert np.array_equal(d, O.sssp(g, src)[0])


def test_zero_weights_self_loops_duplicates(gr, gpu_ctx):
    ro = np.array([0, 3, 5, 6, 6, 8], dtype=np.int32)
    ci = np.array([1, 1, 0, 2, 4, 0, 4, 3], dtype=np.int32)
    w = np.array([2.0, 0.5, 1.0, 0.0, 3.0, 1.0, 0.0, 0.25], dtype=np.float32)
    g = O.Csr(ro, ci, w)
    for s in range(5):
        d, _ = run_sssp(gr, gpu_ctx, ro, ci, w, s)
        assert np.array_equal(d, O.sssp(g, s)[0])
    d, _ = run_sssp(gr, gpu_ctx, ro, ci, w, 3)  # vertex 3 has no out-edges
    assert d[3] == 0 and np.all(np.delete(d, 3) == FMAX)  # unreached = FLT_MAX, not inf (sssp.hxx:72-73)


def test_near_far_schedule_equals_plain_and_does_less_work(gr, gpu_ctx):
    """Delta-stepping (near-far) and plain label-correcting must give the same bits; on a
    weighted lattice the near-far schedule must relax far fewer edges."""
    _, c = gr.generate("road", 300 * 300, a=0.75, c=1.0, seed=3)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = 150 * 300 + 150
    want = O.sssp(g, src)[0]
    # (this lattice is big enough for the block-asynchronous path, the default for road-like graphs: ask for near-far)
    d_nf, st_nf = run_sssp(gr, gpu_ctx, g.row_offsets, g.column_indices, g.values, src,
                           gr.options_t(engine_flags=gr.FLAG_SSSP_NEAR_FAR))
    d_pl, st_pl = run_sssp(gr, gpu_ctx, g.row_offsets, g.column_indices, g.values, src, gr.options_t(engine_flags=0x10))
    assert np.array_equal(d_nf, want) and np.array_equal(d_pl, want)
    assert st_nf["aux"] > 1  # several buckets were opened
    assert st_nf["edges_visited"] < st_pl["edges_visited"] / 3
    # heavy-tailed weights: most buckets empty -> the bucket jump must skip them
    w = c.nonzero_values.copy()
    rng = np.random.default_rng(0)
    w[rng.random(len(w)) < 0.01] = 1.0e6
    g2 = O.Csr(c.row_offsets, c.column_indices, w)
    d2, _ = run_sssp(gr, gpu_ctx, g2.row_offsets, g2.column_indices, w, src)
    assert np.array_equal(d2, O.sssp(g2, src)[0])
    # tiny fractional weights next to huge ones (fp guard of the bucket bound)
    w3 = (rng.random(len(w), dtype=np.float32) * np.float32(1e-3) + np.float32(1e-6)).astype(np.float32)
    w3[rng.random(len(w)) < 0.02] = np.float32(3.0e7)
    g3 = O.Csr(c.row_offsets, c.column_indices, w3)
    d3, _ = run_sssp(gr, gpu_ctx, g3.row_offsets, g3.column_indices, w3, src)
    assert np.array_equal(d3, O.sssp(g3, src)[0])


def test_medium_weighted_rmat(gr, gpu_ctx):
    _, c = gr.generate("rmat", 1 << 18, 3_000_000, seed=9)
    rng = np.random.default_rng(1)
    w = rng.integers(1, 1001, c.number_of_nonzeros).astype(np.float32)
    g = O.Csr(c.row_offsets, c.column_indices, w)
    src = int(np.argmax(np.diff(g.row_offsets)))
    d, st = run_sssp(gr, gpu_ctx, g.row_offsets, g.column_indices, w, src)
    assert np.array_equal(d, O.sssp(g, src)[0])
    assert O.check_sssp(g, src, d) == 0


def test_full_size_road_standin_properties(gr, gpu_ctx, monkeypatch):
    """BASELINE.json configs[2] size: 4894x4894 lattice (23,951,236 V, ~57.7 M E),
    weighted variant U{1..1000}; exact fixed-point characterisation by the oracle."""
    _, c = gr.generate("road", 4894 * 4894, a=0.602, c=1.0, seed=42)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    assert 55_000_000 < g.n_edges < 60_000_000
    src = (4894 // 2) * 4894 + 4894 // 2
    # the near-far schedule; on a library built with the block-asynchronous relaxation (grx_block.hip: libgrx_block.so,
    # GRX_BLOCK=1) that path too -- GRX_FLAG_NO_BLOCK_ASYNC then selects near-far
    blocky = gr.has_block_async()
    if blocky:
        monkeypatch.setenv("GRX_BLOCK", "1")
    got = []
    for o in (None, gr.options_t(engine_flags=gr.FLAG_NO_BLOCK_ASYNC)):
        d, st = run_sssp(gr, gpu_ctx, g.row_offsets, g.column_indices, g.values, src, o)
        assert (d < FMAX).sum() > g.n_vertices // 2
        assert O.check_sssp(g, src, d) == 0
        assert (gr.block_stats(gpu_ctx)["supersteps"] > 0) == (o is None and blocky)
        got.append(d)
    assert np.array_equal(got[0], got[1])


def test_full_size_road_standin_unit_weights(gr, gpu_ctx, monkeypatch):
    """BASELINE.json configs[2] as the reference loader produces it from the pattern file road_usa.mtx:
    every weight 1.0 (io/matrix_market.hxx:170-171).  Full size; distances must equal the BFS depths of
    the same run (bit-exact, as floats) and pass the oracle's exact fixed-point check."""
    import torch
    props, c = gr.generate("road", 4894 * 4894, a=0.602, c=0.0, seed=42)
    assert np.all(c.nonzero_values == 1.0)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = (4894 // 2) * 4894 + 4894 // 2
    depths, _, ev = O.bfs_queue(g, src)
    reached = depths != np.iinfo(np.int32).max
    if gr.has_block_async():
        monkeypatch.setenv("GRX_BLOCK", "1")
    # default: the BFS engine (block-asynchronous on this road-like graph with GRX_BLOCK=1 on libgrx_block.so) + one pass depths -> distances;
    # GRX_FLAG_SSSP_NO_BFS: the SSSP paths (block-asynchronous with the weight array; with GRX_FLAG_NO_BLOCK_ASYNC the
    # level-synchronous relaxation kernels of grx_sssp.hip); GRX_FLAG_NO_BLOCK_ASYNC alone: the level-synchronous BFS engine
    for flags in (0, gr.FLAG_SSSP_NO_BFS, gr.FLAG_SSSP_NO_BFS | gr.FLAG_NO_BLOCK_ASYNC, gr.FLAG_NO_BLOCK_ASYNC):
        d, st = run_sssp(gr, gpu_ctx, g.row_offsets, g.column_indices, g.values, src, gr.options_t(engine_flags=flags))
        assert O.check_sssp(g, src, d) == 0
        assert np.array_equal(d[reached], depths[reached].astype(np.float32)), flags
        assert np.all(d[~reached] == FMAX)
        if flags != gr.FLAG_SSSP_NO_BFS:  # (the weighted block schedule reports every relaxation, re-relaxations included)
            assert st["edges_visited"] == ev, flags  # every reached vertex's out-edges exactly once
        else:
            assert st["edges_visited"] >= ev


def test_regression_late_workgroups_of_a_multi_level_launch(gr, gpu_ctx):
    """The many-levels-per-launch body (grx_mid.hpp) rewrites the control block while workgroups of the same grid
    may still be STARTING; they must find ctrl.mode == 3 and leave.  (With `mode = 0` written at the hand-back a
    late workgroup ran the per-level advance on the half-rewritten control block: a sporadic memory fault, first
    seen on a near-far search from an isolated source right after another near-far search.)"""
    import torch
    rng = np.random.default_rng(5)
    for _ in range(2):
        V = int(rng.integers(200, 30000))
        E = int(rng.integers(V, 10 * V))
    _, c = gr.generate("rmat", V, E, seed=501)
    w = (rng.random(E, dtype=np.float32) * np.float32(9.7) + np.float32(0.01)).astype(np.float32)
    g = O.Csr(c.row_offsets, c.column_indices, w)
    deg = np.diff(g.row_offsets)
    hub, lone = int(np.argmax(deg)), int(np.nonzero(deg == 0)[0][0])
    csr = gr.csr_t.from_arrays(g.row_offsets, g.column_indices, w)
    G = gr.build_graph(gr.graph_properties_t(True, True, False), csr, gpu_ctx)
    d = torch.empty(V, dtype=torch.float32, device="cuda:0")
    want = {s: O.sssp(g, s)[0] for s in (hub, lone)}
    nf, plain = gr.options_t(engine_flags=0x20), gr.options_t(engine_flags=0x10)
    for rep in range(8):
        for s, o in ((hub, nf), (lone, nf), (hub, plain), (lone, nf), (hub, None), (lone, plain)):
            gr.sssp(G, s, d, None, gpu_ctx, o)
            assert np.array_equal(d.cpu().numpy(), want[s]), (rep, s)
