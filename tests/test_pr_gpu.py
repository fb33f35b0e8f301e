"""PageRank parity on the GPU through the C ABI.

The reference pins nothing for PR (no CPU oracle, no test); its own GPU result
depends on atomicAdd order.  Tolerances (north_star + SURVEY 8c): against the
float64 evaluation of the same recurrence run for the SAME number of
iterations, |delta| <= 1e-6 absolute and <= 1e-4 relative; the iteration count
must equal the float64 recurrence's (+-1 where the convergence test sits within
rounding of the threshold).  The sequential fp32 restatement is NOT the yardstick
for the count on hub-heavy graphs: summing 1e5 tiny terms into one fp32
accumulator leaves ~1e-5 absolute noise at hubs, above tol=1e-6, so it (like the
reference's atomicAdd order) needs 16 iterations where float64 needs 9 on the
kron stand-in; the pull kernel's blocked sums track float64."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
ABS_TOL = 1e-6
REL_TOL = 1e-4


def run_pr(gr, ctx, g, alpha=0.85, tol=1e-6, weighted=True, max_iterations=0, engine_flags=0):
    import torch
    csr = gr.csr_t.from_arrays(g.row_offsets, g.column_indices, g.values)
    G = gr.build_graph(gr.graph_properties_t(True, weighted, False), csr, ctx)
    p = torch.zeros(g.n_vertices, dtype=torch.float32, device="cuda:0")
    res = gr.pr_result_t(p)
    ms = gr.pr_run(G, gr.pr_param_t(alpha, tol, gr.options_t(max_iterations=max_iterations,
                                                            engine_flags=engine_flags)), res, ctx)
    assert ms >= 0
    return p.cpu().numpy(), res.iterations


def check(g, p, it, alpha=0.85, tol=1e-6):
    _, it64, _ = O.pr_f64(g, alpha, tol)
    assert abs(it - it64) <= 1, (it, it64)
    p64, _, _ = O.pr_f64(g, alpha, tol, force_iterations=it)
    diff = np.abs(p.astype(np.float64) - p64)
    assert diff.max() <= ABS_TOL, diff.max()
    rel = diff / np.maximum(np.abs(p64), 1e-30)
    assert rel.max() <= REL_TOL, rel.max()
    assert abs(float(p.astype(np.float64).sum()) - 1.0) < 1e-3


def test_chesapeake(gr, gpu_ctx, golden):
    g = O.Csr(golden["chesapeake_ro"], golden["chesapeake_ci"], golden["chesapeake_w"])
    p, it = run_pr(gr, gpu_ctx, g)
    check(g, p, it)
    assert it == O.pr_f32(g)[1]


def test_dangling_and_weighted(gr, gpu_ctx, golden):
    # directed R-MAT with sinks: dangling mass is redistributed every iteration (pr.hxx:125-134)
    g = O.Csr(golden["rmat_ro"], golden["rmat_ci"], np.ones(len(golden["rmat_ci"]), np.float32))
    assert (np.diff(g.row_offsets) == 0).any()
    p, it = run_pr(gr, gpu_ctx, g)
    check(g, p, it)
    r = O.Csr(golden["road_ro"], golden["road_ci"], golden["road_w"])
    p, it = run_pr(gr, gpu_ctx, r)
    check(r, p, it)
    t = O.Csr(golden["tiny_ro"], golden["tiny_ci"], golden["tiny_w"])
    p, it = run_pr(gr, gpu_ctx, t, tol=1e-7)
    check(t, p, it, tol=1e-7)


def test_alpha_tol_variants_and_iteration_cap(gr, gpu_ctx, golden):
    g = O.Csr(golden["rmat_ro"], golden["rmat_ci"], np.ones(len(golden["rmat_ci"]), np.float32))
    for alpha, tol in ((0.5, 1e-6), (0.95, 1e-5), (0.85, 1e-8)):
        p, it = run_pr(gr, gpu_ctx, g, alpha, tol)
        check(g, p, it, alpha, tol)
    p, it = run_pr(gr, gpu_ctx, g, max_iterations=3)
    assert it == 3
    p64, _, _ = O.pr_f64(g, force_iterations=3)
    assert np.abs(p - p64).max() <= ABS_TOL


def test_xcd_blocked_layout_equals_plain_layout(gr, gpu_ctx, golden):
    """The XCD-blocked pull (in-edges bucketed by source block, 8 partial sums per row) is a
    different summation order only: same iteration count, ranks within fp32 rounding."""
    graphs = [O.Csr(golden["rmat_ro"], golden["rmat_ci"], np.ones(len(golden["rmat_ci"]), np.float32)),
              O.Csr(golden["road_ro"], golden["road_ci"], golden["road_w"])]
    _, c = gr.generate("rmat_sym", 1 << 15, 600_000, seed=13)  # hubs => long-row pieces in every bucket
    graphs.append(O.Csr(c.row_offsets, c.column_indices, c.nonzero_values))
    for g in graphs:
        p_plain, it_plain = run_pr(gr, gpu_ctx, g, engine_flags=0x40)
        p_xcd, it_xcd = run_pr(gr, gpu_ctx, g, engine_flags=0x80)
        assert it_plain == it_xcd
        assert np.abs(p_plain.astype(np.float64) - p_xcd).max() <= 2e-7 * max(1.0, float(p_plain.max()) * 1e3)
        check(g, p_xcd, it_xcd)


def test_hub_rows_are_split(gr, gpu_ctx):
    # symmetric R-MAT: in-degree hubs far beyond one workgroup's 2048-nnz block
    _, c = gr.generate("rmat_sym", 1 << 16, 1_500_000, seed=3)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    assert np.diff(g.row_offsets).max() > 5000
    p, it = run_pr(gr, gpu_ctx, g, weighted=False)
    check(g, p, it)


def test_kron_scale_standin(gr, gpu_ctx):
    """BASELINE.json configs[3] (kron_g500-logn21) at reduced edge count so the CPU
    float64 yardstick stays in seconds: 2^21 vertices, ~40 M edges."""
    _, c = gr.generate("rmat_sym", 1 << 21, 20_000_000, seed=42)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    p, it = run_pr(gr, gpu_ctx, g, weighted=False)
    check(g, p, it)
