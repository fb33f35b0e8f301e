"""Binned relaxation (grx_relax.hpp): the fat levels of a weighted SSSP on a dense graph as scatter + sweep with the
minimum taken in LDS.  Distances must stay bit-identical (fp32 ==) to the oracle's Dijkstra and to the relax-per-edge
kernels (GRX_FLAG_SSSP_NO_BINS) -- the fixed point of sssp.hxx:116-130 does not depend on the schedule."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _run(gr, ctx, c, w, src, flags=0):
    import torch
    csr = gr.csr_t.from_arrays(c.row_offsets, c.column_indices, w)
    G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=True, symmetric=False), csr, ctx)
    d = torch.full((G.get_number_of_vertices(),), -1.0, dtype=torch.float32, device="cuda:0")
    gr.sssp(G, src, d, None, ctx, gr.options_t(engine_flags=flags))
    st = gr.run_stats(ctx)
    gr.sssp(G, src, d, None, ctx, gr.options_t(engine_flags=flags | gr.FLAG_PROFILE))
    prof = gr.level_profile(ctx)
    return d.cpu().numpy(), st, prof


def _weights(kind, n, rng):
    if kind == "int":
        return rng.integers(1, 1001, n).astype(np.float32)
    return (rng.random(n, dtype=np.float32) * np.float32(9.7) + np.float32(0.01)).astype(np.float32)


@pytest.mark.parametrize("kind", ["int", "float"])
def test_binned_levels_equal_oracle_and_plain_kernels(gr, gpu_ctx, monkeypatch, kind):
    # small thresholds: graphs the oracle finishes in seconds still get binned levels
    monkeypatch.setenv("GRX_RBIN_MIN_GRAPH_EDGES", "0")
    monkeypatch.setenv("GRX_RBIN_MIN_EDGES", "20000")
    rng = np.random.default_rng(11)
    binned_levels = 0
    for trial, (V, E) in enumerate([(1 << 16, 1_500_000), (70_001, 900_000), (1 << 18, 3_000_000), (3000, 60_000)]):
        _, c = gr.generate("rmat", V, E, seed=700 + trial)
        w = _weights(kind, c.number_of_nonzeros, rng)
        g = O.Csr(c.row_offsets, c.column_indices, w)
        for src in (int(np.argmax(np.diff(g.row_offsets))), int(rng.integers(0, V))):
            want = O.sssp(g, src)[0]
            d, st, prof = _run(gr, gpu_ctx, c, w, src)
            assert np.array_equal(d, want), (kind, V, E, src)
            d0, st0, _ = _run(gr, gpu_ctx, c, w, src, gr.FLAG_SSSP_NO_BINS)
            assert np.array_equal(d0, want)
            binned_levels += sum(1 for r in prof if r["bottom_up"] == 2)
    assert binned_levels >= 6  # the path under test was taken


def test_hub_bins_are_relaxed_in_parts(gr, gpu_ctx, monkeypatch):
    """many parts per bin: the parts of a bin race on its labels with device-scope atomics and agree through the stamp on who
    emits an improved vertex -- no vertex may enter the next frontier twice (edges relaxed == the plain kernels' per level
    would differ otherwise) and the distances stay exact"""
    monkeypatch.setenv("GRX_RBIN_MIN_GRAPH_EDGES", "0")
    monkeypatch.setenv("GRX_RBIN_MIN_EDGES", "100000")
    monkeypatch.setenv("GRX_RBIN_PARTS", "4096")
    rng = np.random.default_rng(3)
    _, c = gr.generate("rmat", 1 << 17, 4_000_000, seed=31)
    w = _weights("int", c.number_of_nonzeros, rng)
    g = O.Csr(c.row_offsets, c.column_indices, w)
    src = int(np.argmax(np.diff(g.row_offsets)))
    want = O.sssp(g, src)[0]
    d, st, prof = _run(gr, gpu_ctx, c, w, src)
    assert np.array_equal(d, want)
    assert sum(1 for r in prof if r["bottom_up"] == 2) >= 2
    deg = np.diff(g.row_offsets)
    for r in prof:  # a frontier holds every vertex at most once: its edges never exceed the graph's
        assert r["frontier_size"] <= g.n_vertices and r["edges"] <= g.n_edges
    assert O.check_sssp(g, src, d) == 0


def test_scatter_safety_net(gr, monkeypatch):
    """one XCD's scatter workgroups drop out: the sweep's coverage check stops the search, the host repeats it with static units"""
    monkeypatch.setenv("GRX_RBIN_MIN_GRAPH_EDGES", "0")
    monkeypatch.setenv("GRX_RBIN_MIN_EDGES", "20000")
    monkeypatch.setenv("GRX_SC2_FAULT_XCD", "2")
    ctx = gr.multi_context_t(0)  # a fresh context: the static mode is sticky per context
    rng = np.random.default_rng(5)
    _, c = gr.generate("rmat", 1 << 16, 1_200_000, seed=77)
    w = _weights("int", c.number_of_nonzeros, rng)
    g = O.Csr(c.row_offsets, c.column_indices, w)
    src = int(np.argmax(np.diff(g.row_offsets)))
    d, st, prof = _run(gr, ctx, c, w, src)
    assert np.array_equal(d, O.sssp(g, src)[0])


def test_first_batch_follows_the_previous_search(gr, gpu_ctx, monkeypatch):
    """Weighted SSSP on a dense graph: the first blind batch of launch groups is the previous search's level count + 1
    (grx_sssp.hip) -- alternating sources whose searches need different numbers of levels (a tail hung onto the graph makes one
    much deeper), with and without the prediction, must give the oracle's distances."""
    import torch
    monkeypatch.setenv("GRX_RBIN_MIN_GRAPH_EDGES", "0")
    monkeypatch.setenv("GRX_RBIN_MIN_EDGES", "20000")
    rng = np.random.default_rng(19)
    _, c = gr.generate("rmat", 1 << 16, 1_500_000, seed=41)
    V, L = c.number_of_rows, 40
    deg = np.diff(c.row_offsets)
    hub = int(np.argmax(deg))
    # tail: V -> V+1 -> ... -> V+L-1 -> hub (directed towards the graph); the graph stays dense enough (E >= 8 V) for the
    # prediction to apply
    src_e = np.concatenate([np.repeat(np.arange(V, dtype=np.int64), deg), V + np.arange(L)])
    dst_e = np.concatenate([c.column_indices.astype(np.int64), np.concatenate([V + 1 + np.arange(L - 1), [hub]])])
    order = np.argsort(src_e, kind="stable")
    ro = np.zeros(V + L + 1, np.int64)
    np.cumsum(np.bincount(src_e, minlength=V + L), out=ro[1:])
    ci = dst_e[order].astype(np.int32)
    w = _weights("int", len(ci), rng)
    g = O.Csr(ro.astype(np.int32), ci, w)
    assert g.n_edges >= 8 * g.n_vertices
    G = gr.build_graph(gr.graph_properties_t(directed=True, weighted=True, symmetric=False),
                       gr.csr_t.from_arrays(g.row_offsets, g.column_indices, w), gpu_ctx)
    want = {s: O.sssp(g, s)[0] for s in (hub, V)}
    d = torch.empty(g.n_vertices, dtype=torch.float32, device="cuda:0")
    depths = {}
    for hint in ("1", "0"):
        monkeypatch.setenv("GRX_GROUP_HINT", hint)
        for s in (hub, hub, V, V, hub, V, hub):
            d.fill_(-1.0)
            gr.sssp(G, s, d, None, gpu_ctx, gr.options_t())
            assert np.array_equal(d.cpu().numpy(), want[s]), (hint, s)
            depths[s] = gr.run_stats(gpu_ctx)["search_depth"]
    assert depths[V] >= depths[hub] + L - 2, depths  # the tail source IS deeper: its prediction from the hub falls short
