"""The many-levels-per-launch body (gunrock_amd/csrc/grx_mid.hpp) on the paths a benchmark graph never takes:
the shared overflow area behind the private output regions, the hand-back of a growing frontier to the regular
kernels, the launch-pair-per-level schedule.  Test knobs (read per run by the library): GRX_MID_SEG_CAP = entries of a
private region in use, GRX_MID_EXIT_V = frontier size at which the body hands back, GRX_MID=0, GRX_NF_FOLD = far-pile size up to which a near-far search
changes bucket inside the launch.
BFS depths / SSSP distances must equal the oracle's bit for bit in every configuration
(what the reference's --validate checks: examples/algorithms/bfs/bfs.cu:96-113, sssp/sssp.cu)."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
KNOBS = ("GRX_MID_SEG_CAP", "GRX_MID_EXIT_V", "GRX_MID_VERSION", "GRX_MID", "GRX_NF_FOLD")
CONFIGS = (
    {},                                                   # defaults
    {"GRX_MID_SEG_CAP": "256"},                           # a workgroup's second flush of a level overflows
    {"GRX_MID_SEG_CAP": "0"},                             # everything goes through the overflow area
    {"GRX_MID_EXIT_V": "3000"},                           # early hand-back (regions become tiles)
    {"GRX_MID_SEG_CAP": "256", "GRX_MID_EXIT_V": "9000"}, # hand-back with a non-empty overflow area
    {"GRX_MID": "0"},                                     # one launch pair per level
    # near-far SSSP (round 6: the next bucket is pulled out of the far pile inside the launch while the pile is small enough)
    {"GRX_NF_FOLD": "0"},                                 # every bucket change through the head kernel, as before round 6
    {"GRX_NF_FOLD": "300"},                               # small piles inside the launch, larger ones through the head
    {"GRX_NF_FOLD": "65536", "GRX_MID_SEG_CAP": "0"},     # the pulled bucket goes through the overflow area
    {"GRX_NF_FOLD": "65536", "GRX_MID_SEG_CAP": "256", "GRX_MID_EXIT_V": "3000"},  # ... and is handed back as tiles
)


@pytest.fixture(autouse=True)
def _clean_env():
    saved = {k: os.environ.pop(k, None) for k in KNOBS}
    yield
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def sparse_random_graph(n, out_degree, seed):
    """every vertex points at `out_degree` uniformly random vertices: the BFS frontier grows by that factor per
    level until it saturates -- it passes through the size window of the multi-level body on the way up"""
    rng = np.random.default_rng(seed)
    ro = (np.arange(n + 1, dtype=np.int64) * out_degree).astype(np.int32)
    ci = rng.integers(0, n, n * out_degree).astype(np.int32)
    return ro, ci


def lattice(gr, side, weighted, seed):
    _, c = gr.generate("road", side * side, a=0.62, c=1.0 if weighted else 0.0, seed=seed)
    return c.row_offsets, c.column_indices, c.nonzero_values


def run_bfs(gr, ctx, ro, ci, src, direction):
    import torch
    G = gr.build_graph(gr.graph_properties_t(True, False, False), gr.csr_t.from_arrays(ro, ci), ctx)
    d = torch.empty(len(ro) - 1, dtype=torch.int32, device="cuda:0")
    # (these graphs are road-like: by default they would be searched block-asynchronously, grx_block.hip -- this file is
    # about the level-synchronous multi-level body)
    o = gr.options_t(advance_load_balance=gr.merge_path, advance_direction=direction, engine_flags=gr.FLAG_NO_BLOCK_ASYNC)
    out = []
    for env in CONFIGS:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        gr.bfs(G, src, d, None, ctx, o)
        out.append((env, d.cpu().numpy().copy(), gr.run_stats(ctx)))
    return out


def run_sssp(gr, ctx, ro, ci, w, src, flags=0):
    import torch
    G = gr.build_graph(gr.graph_properties_t(True, True, False), gr.csr_t.from_arrays(ro, ci, w), ctx)
    d = torch.empty(len(ro) - 1, dtype=torch.float32, device="cuda:0")
    out = []
    for env in CONFIGS:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        gr.sssp(G, src, d, None, ctx, gr.options_t(engine_flags=flags | gr.FLAG_NO_BLOCK_ASYNC))
        out.append((env, d.cpu().numpy().copy(), gr.run_stats(ctx)))
    return out


def test_growing_frontier_bfs_and_unit_sssp(gr, gpu_ctx):
    ro, ci = sparse_random_graph(400_000, 3, seed=11)
    g = O.Csr(ro, ci, np.ones(len(ci), dtype=np.float32))
    for src in (0, 123_457):
        want, _, ev = O.bfs_queue(g, src)
        for env, d, st in run_bfs(gr, gpu_ctx, ro, ci, src, gr.forward):
            assert np.array_equal(d, want), env
            assert st["edges_visited"] == ev, env
        fwant = want.astype(np.float64)
        fwant[want == np.iinfo(np.int32).max] = np.finfo(np.float32).max
        # unit weights: the BFS engine + depths -> distances (default), and the SSSP relaxation kernels' own multi-level body
        for flags in (0, gr.FLAG_SSSP_NO_BFS):
            for env, d, st in run_sssp(gr, gpu_ctx, ro, ci, g.values, src, flags):
                assert np.array_equal(d, fwant.astype(np.float32)), (flags, env)


def test_lattice_all_paths(gr, gpu_ctx):
    side = 700
    ro, ci, w = lattice(gr, side, weighted=False, seed=3)
    g = O.Csr(ro, ci, w)
    src = (side // 2) * side + side // 2
    want, _, ev = O.bfs_queue(g, src)
    for env, d, st in run_bfs(gr, gpu_ctx, ro, ci, src, gr.forward):
        assert np.array_equal(d, want), env
        assert st["edges_visited"] == ev, env
    for env, d, st in run_bfs(gr, gpu_ctx, ro, ci, src, gr.optimized):
        assert np.array_equal(d, want), env


def test_weighted_lattice_near_far_and_plain(gr, gpu_ctx):
    side = 500
    ro, ci, w = lattice(gr, side, weighted=True, seed=9)
    g = O.Csr(ro, ci, w)
    src = (side // 2) * side + side // 2
    want = O.sssp(g, src)[0]
    for flags in (0, 0x20, 0x10):  # schedule chosen by the engine, near-far forced, plain label-correcting forced
        for env, d, st in run_sssp(gr, gpu_ctx, ro, ci, w, src, flags):
            assert np.array_equal(d, want), (flags, env)


def test_weighted_sparse_random_graph(gr, gpu_ctx):
    ro, ci = sparse_random_graph(300_000, 3, seed=5)
    rng = np.random.default_rng(6)
    w = rng.integers(1, 64, len(ci)).astype(np.float32)
    g = O.Csr(ro, ci, w)
    want = O.sssp(g, 7)[0]
    for flags in (0, 0x20):
        for env, d, st in run_sssp(gr, gpu_ctx, ro, ci, w, 7, flags):
            assert np.array_equal(d, want), (flags, env)
