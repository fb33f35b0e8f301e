"""Block-asynchronous relaxation on the GPU (gunrock_amd/csrc/grx_block.hip): road-like graphs (fewer than 4 edges per
vertex, >= 65536 vertices) take this path by default for BFS, unit-weight SSSP (through the BFS engine) and weighted SSSP.
Depths / distances must equal the oracle's bit for bit (bfs_cpu.hxx:32-63, sssp_cpu.hxx:36-67) for every block size and
bucket width, from the centre, a corner and an isolated vertex, repeatedly, with and without GRX_FLAG_PROFILE; the
level-synchronous kernels (GRX_FLAG_NO_BLOCK_ASYNC) must agree; the counters keep the reference's meaning."""
import os

import numpy as np
import pytest

import oracle_lib as O

def _has_block():
    try:
        import torch  # noqa: F401  (load order: torch's HIP runtime before libgrx's -- the other way round no device is found)
        import gunrock_amd as gr
        return gr.has_block_async()
    except Exception:  # noqa: BLE001
        return False


# (the default library does not carry this path since round 6: tests/test_block_variant.py runs this file against libgrx_block.so)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _has_block(), reason="library built without grx_block.hip (python -m gunrock_amd.build --with-block)")]
INF = np.iinfo(np.int32).max
KNOBS = ("GRX_BLOCK_NV", "GRX_BLOCK_NV_W", "GRX_BLOCK_DELTA", "GRX_BLOCK_DELTA_W", "GRX_BLOCK", "GRX_BLOCK_WG_PER_CU")


@pytest.fixture(autouse=True)
def _clean_env():
    saved = {k: os.environ.pop(k, None) for k in KNOBS}
    os.environ["GRX_BLOCK"] = "1"  # (whatever the library's default is: this file is about that path)
    yield
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def _lattice(gr, side, weighted, seed):
    _, c = gr.generate("road", side * side, a=0.602, c=1.0 if weighted else 0.0, seed=seed)
    return O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)


def test_bfs_lattice_every_block_size_and_bucket(gr, gpu_ctx):
    import torch
    side = 600
    g = _lattice(gr, side, False, 3)
    deg = np.diff(g.row_offsets)
    sources = [(side // 2) * side + side // 2, 0, int(np.nonzero(deg == 0)[0][0])]
    want = {s: O.bfs_queue(g, s) for s in sources}
    d = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
    for env in ({}, {"GRX_BLOCK_NV": "2048"}, {"GRX_BLOCK_NV": "8192"}, {"GRX_BLOCK_DELTA": "1"}, {"GRX_BLOCK_DELTA": "37"},
                {"GRX_BLOCK_DELTA": "100000"}, {"GRX_BLOCK_WG_PER_CU": "1"}):
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(dict(env, GRX_BLOCK="1"))
        # a fresh handle per configuration: the block structure is cached in it
        G = gr.build_graph(gr.graph_properties_t(True, False, False), gr.csr_t.from_arrays(g.row_offsets, g.column_indices), gpu_ctx)
        for rep in range(2):
            for s in sources:
                for flags in (0, gr.FLAG_PROFILE):
                    gr.bfs(G, s, d, None, gpu_ctx, gr.options_t(engine_flags=flags))
                    depths, _, ev = want[s]
                    assert np.array_equal(d.cpu().numpy(), depths), (env, s, flags)
                    st, bs = gr.run_stats(gpu_ctx), gr.block_stats(gpu_ctx)
                    assert bs["supersteps"] >= 1 and bs["blocks"] * bs["block_vertices"] >= g.n_vertices, (env, bs)
                    assert st["edges_visited"] == ev and st["vertices_visited"] == int((depths != INF).sum())
                    assert st["search_depth"] == int(depths[depths != INF].max()) + 1
                    assert bs["edges_relaxed"] >= ev
        # the level-synchronous kernels on the same handle
        gr.bfs(G, sources[0], d, None, gpu_ctx, gr.options_t(engine_flags=gr.FLAG_NO_BLOCK_ASYNC))
        assert np.array_equal(d.cpu().numpy(), want[sources[0]][0]) and gr.block_stats(gpu_ctx)["supersteps"] == 0
        del G


def test_weighted_and_unit_sssp_lattice(gr, gpu_ctx):
    import torch
    side = 500
    for weighted in (True, False):
        g = _lattice(gr, side, weighted, 9)
        deg = np.diff(g.row_offsets)
        sources = [(side // 2) * side + side // 2, side * side - 1, int(np.nonzero(deg == 0)[0][0])]
        want = {s: O.sssp(g, s)[0] for s in sources}
        d = torch.empty(g.n_vertices, dtype=torch.float32, device="cuda:0")
        for env in ({}, {"GRX_BLOCK_NV_W": "4096", "GRX_BLOCK_NV": "8192"}, {"GRX_BLOCK_DELTA_W": "1", "GRX_BLOCK_DELTA": "3"},
                    {"GRX_BLOCK_DELTA_W": "100000"}):
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(dict(env, GRX_BLOCK="1"))
            G = gr.build_graph(gr.graph_properties_t(True, True, False),
                               gr.csr_t.from_arrays(g.row_offsets, g.column_indices, g.values), gpu_ctx)
            for s in sources:
                for flags in (0, gr.FLAG_PROFILE, gr.FLAG_SSSP_NO_BFS):
                    gr.sssp(G, s, d, None, gpu_ctx, gr.options_t(engine_flags=flags))
                    assert np.array_equal(d.cpu().numpy(), want[s]), (weighted, env, s, flags)
                    assert gr.block_stats(gpu_ctx)["supersteps"] >= 1
            gr.sssp(G, sources[0], d, None, gpu_ctx, gr.options_t(engine_flags=gr.FLAG_NO_BLOCK_ASYNC))
            assert np.array_equal(d.cpu().numpy(), want[sources[0]]) and gr.block_stats(gpu_ctx)["supersteps"] == 0
            del G


def test_sparse_random_graph_without_locality_and_fractional_weights(gr, gpu_ctx):
    """no spatial locality at all (most edges leave their block) and weights whose sums are not exact in fp32: slow for
    this schedule, but the fixed point is the same"""
    import torch
    rng = np.random.default_rng(11)
    n, k = 120_000, 3
    ro = (np.arange(n + 1, dtype=np.int64) * k).astype(np.int32)
    ci = rng.integers(0, n, n * k).astype(np.int32)
    w = (rng.random(n * k, dtype=np.float32) * np.float32(9.7) + np.float32(0.01)).astype(np.float32)
    g = O.Csr(ro, ci, w)
    G = gr.build_graph(gr.graph_properties_t(True, True, False), gr.csr_t.from_arrays(ro, ci, w), gpu_ctx)
    d = torch.empty(n, dtype=torch.float32, device="cuda:0")
    for s in (0, 77_777):
        gr.sssp(G, s, d, None, gpu_ctx)
        assert np.array_equal(d.cpu().numpy(), O.sssp(g, s)[0])
        assert gr.block_stats(gpu_ctx)["supersteps"] >= 1
    di = torch.empty(n, dtype=torch.int32, device="cuda:0")
    Gu = gr.build_graph(gr.graph_properties_t(True, False, False), gr.csr_t.from_arrays(ro, ci), gpu_ctx)
    gr.bfs(Gu, 5, di, None, gpu_ctx)
    assert np.array_equal(di.cpu().numpy(), O.bfs_queue(O.Csr(ro, ci, np.ones(len(ci), np.float32)), 5)[0])
