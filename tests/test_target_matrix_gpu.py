"""north_star's target matrix at FULL size: {BFS, SSSP, PageRank} x {soc-LiveJournal1, kron_g500-logn21} stand-ins
(SURVEY 8d C2' / C4'), every cell compared with the oracle's own arrays:
  BFS   depths  == orc_bfs_queue                          (bfs.hxx:105-146; bit-exact)
  SSSP  distances == orc_sssp (priority-queue Dijkstra)   (sssp.hxx:104-159; fp32 ==), unit weights -- what the reference
        loader makes of these pattern files (io/matrix_market.hxx:170-171) -- and U{1..1000}
  PR    |ours - float64 recurrence| <= 1e-6 at equal iteration count, iteration count == float64's (+-1), and -- when the
        reference compiled here travels with the tree -- against the reference's own GPU PageRank at equal iteration count
        (pr.hxx:107-195).
Also here: uniform weights other than 1.0 (the k-fold fp32 sum, its saturation at FLT_MAX), and a deterministic read of
the labels right behind a GRX_FLAG_ASYNC_RETURN search."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max
INF = np.iinfo(np.int32).max
_cache = {}


def _graph(gr, name):
    """the bench stand-ins, generated once per session: (properties, csr, oracle Csr, source)"""
    if name not in _cache:
        import bench
        props, csr, src, _ = bench.load_workload(gr, name)
        _cache.clear()  # one full-size graph at a time on the host
        _cache[name] = (props, csr, O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values), src)
    return _cache[name]


def _sssp(gr, ctx, props, csr, src, options=None):
    import torch
    G = gr.build_graph(props, csr, ctx)
    d = torch.full((G.get_number_of_vertices(),), -1.0, dtype=torch.float32, device="cuda:0")
    gr.sssp(G, src, d, None, ctx, options)
    return d.cpu().numpy(), gr.run_stats(ctx)


def _sssp_cells(gr, ctx, name):
    import bench
    import copy
    props, csr, g, src = _graph(gr, name)
    assert np.all(csr.nonzero_values == 1.0)  # pattern stand-in: unit weights, like the published file
    # unit weights: default (BFS engine), direction-optimising BFS engine, and the relaxation kernels
    want, _ = O.sssp(g, src)
    for o in (None, gr.options_t(advance_direction=gr.optimized), gr.options_t(engine_flags=gr.FLAG_SSSP_NO_BFS)):
        d, st = _sssp(gr, ctx, props, csr, src, o)
        assert np.array_equal(d, want), (name, "unit", o and (o.advance_direction, o.engine_flags))
    depths, _, ev = O.bfs_queue(g, src)
    reached = depths != INF
    assert np.array_equal(want[reached], depths[reached].astype(np.float32)) and np.all(want[~reached] == FMAX)
    assert st["edges_visited"] == ev
    # U{1..1000}
    cw = copy.copy(csr)
    cw.nonzero_values = bench.pair_hash_weights(csr)
    cw._device = None
    pw = copy.copy(props)
    pw.weighted = True
    gw = O.Csr(cw.row_offsets, cw.column_indices, cw.nonzero_values)
    want_w, _ = O.sssp(gw, src)
    d, st = _sssp(gr, ctx, pw, cw, src)
    assert np.array_equal(d, want_w), (name, "weighted")
    assert O.check_sssp(gw, src, d) == 0


def test_full_size_lj_bfs_depths_equal_the_oracle_array(gr, gpu_ctx):
    import torch
    props, csr, g, src = _graph(gr, "lj")
    assert (g.n_vertices, g.n_edges) == (4_847_571, 68_993_773)
    want, _, ev = O.bfs_queue(g, src)
    G = gr.build_graph(props, csr, gpu_ctx)
    d = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
    for o in (gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact),
              gr.options_t(advance_direction=gr.optimized)):
        gr.bfs(G, src, d, None, gpu_ctx, o)
        assert np.array_equal(d.cpu().numpy(), want)
        assert gr.run_stats(gpu_ctx)["edges_visited"] == ev


def test_async_return_labels_are_final_when_the_call_returns(gr, gpu_ctx):
    """GRX_FLAG_ASYNC_RETURN (what bench.py times): the call returns when the device has PUBLISHED the end of the search; the
    labels must be final then.  They are read here with a copy on ANOTHER stream that is not ordered behind the
    engine's (so the no-op level groups still draining on the engine's stream cannot hide a late write), alternating
    sources so that a stale array cannot pass."""
    import torch
    props, csr, g, src = _graph(gr, "lj")
    G = gr.build_graph(props, csr, gpu_ctx)
    d = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda:0")
    host = torch.empty(g.n_vertices, dtype=torch.int32).pin_memory()
    side = torch.cuda.Stream()
    deg = np.diff(g.row_offsets)
    other = int(np.argsort(deg)[-2])
    want = {s: O.bfs_queue(g, s)[0] for s in (src, other)}
    for direction in (gr.forward, gr.optimized):
        o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                         advance_direction=direction, engine_flags=gr.FLAG_ASYNC_RETURN)
        for rep in range(6):
            s = (src, other)[rep & 1]
            gr.bfs(G, s, d, None, gpu_ctx, o)
            with torch.cuda.stream(side):
                host.copy_(d, non_blocking=True)
            side.synchronize()
            assert np.array_equal(host.numpy(), want[s]), (direction, rep)
            gpu_ctx.synchronize()


def test_full_size_lj_sssp(gr, gpu_ctx):
    _sssp_cells(gr, gpu_ctx, "lj")


def test_full_size_lj_pr(gr, gpu_ctx):
    """PageRank on the C2' stand-in (directed, 4.85 M V / 69 M E, many dangling vertices): against the float64 recurrence at
    equal iteration count (1e-6 absolute, 1e-4 relative, count +-1) and against the reference's own GPU path."""
    import torch
    props, csr, g, _ = _graph(gr, "lj")
    G = gr.build_graph(props, csr, gpu_ctx)
    p = torch.zeros(g.n_vertices, dtype=torch.float32, device="cuda:0")
    res = gr.pr_result_t(p)
    gr.pr_run(G, gr.pr_param_t(0.85, 1e-6), res, gpu_ctx)
    it = res.iterations
    mine = p.cpu().numpy()
    cmp = [mine]
    have_ref = O.have_ref_gpu()
    if have_ref:
        with O.RefGpuGraph(g) as R:
            ref, k_ref, _ = R.pr(0.85, 1e-6)
        pk = torch.zeros_like(p)
        rk = gr.pr_result_t(pk)
        gr.pr_run(G, gr.pr_param_t(0.85, 0.0, gr.options_t(max_iterations=k_ref)), rk, gpu_ctx)  # tol 0: exactly k_ref
        assert rk.iterations == k_ref
        cmp += [ref, pk.cpu().numpy()]
    delta, err, _ = O.pr_f64_trace(g, max(it + 2, (k_ref + 1) if have_ref else 0, 24), cmp, pattern=True)
    it64 = O.pr_iterations_from_trace(delta)
    assert it64 is not None and abs(it - it64) <= 1, (it, it64)
    assert float(err[0][it - 1]) <= 1e-6, float(err[0][it - 1])
    _, _, p64 = O.pr_f64_trace(g, it, [], pattern=True, want_final=True)
    rel = np.abs(mine.astype(np.float64) - p64) / np.maximum(p64, 1e-30)
    assert rel.max() <= 1e-4, rel.max()
    assert abs(float(mine.astype(np.float64).sum()) - 1.0) < 1e-3
    if have_ref:
        e_ref = float(err[1][k_ref - 1])   # the reference's own distance from exact arithmetic at its count
        e_ours = float(err[2][k_ref - 1])
        d_k = float(np.abs(cmp[2].astype(np.float64) - ref).max())
        print({"ours_iterations": it, "f64_iterations": it64, "ref_iterations": k_ref, "ours_k_vs_ref_k": d_k,
               "ours_k_vs_f64_k": e_ours, "ref_k_vs_f64_k": e_ref})
        # The reference's OWN iteration count is not stable at this size: its fp32 atomicAdd contributions arrive in a different
        # order every run, and its convergence test sits inside that noise (8 iterations on most runs, 22 observed in round 5
        # with its result 6.8e-5 from exact arithmetic).  Equal counts are required when the reference itself agrees with the
        # float64 recurrence; otherwise the run is recorded above and only OUR iterate at ITS count is held to the tolerance.
        if abs(k_ref - it64) <= 1:
            assert abs(it - k_ref) <= 1, (it, k_ref)
        assert e_ours <= 1e-6
        assert d_k <= 1e-6 + e_ref, (d_k, e_ref)


def test_full_size_kron_sssp(gr, gpu_ctx):
    _sssp_cells(gr, gpu_ctx, "kron")


def test_uniform_weights_other_than_one(gr, gpu_ctx):
    """All weights equal to some w != 1: the BFS engine + the k-fold fp32 sum table must give exactly what the
    relaxation (sssp.hxx:121-126) gives -- including sums that are not multiples of w in fp32, sums that stall and
    sums that overflow (a tentative distance that is not < FLT_MAX never replaces the initial label)."""
    import copy
    _, c = gr.generate("rmat", 1 << 15, 400_000, seed=3)
    n = 3000
    path_ro = np.arange(n + 1, dtype=np.int32)
    path_ro[-1] = n - 1
    path_ci = np.arange(1, n, dtype=np.int32)
    for w in (0.1, 2.5, 1e-3, 0.0, 7.0e37, 1.0):
        for ro, ci, src in ((c.row_offsets, c.column_indices, int(np.argmax(np.diff(c.row_offsets)))),
                            (path_ro, path_ci, 0)):
            vals = np.full(len(ci), w, dtype=np.float32)
            g = O.Csr(ro, ci, vals)
            want, _ = O.sssp(g, src)
            csr = gr.csr_t.from_arrays(ro, ci, vals)
            for o in (None, gr.options_t(engine_flags=gr.FLAG_SSSP_NO_BFS)):
                d, _ = _sssp(gr, gpu_ctx, gr.graph_properties_t(True, True, False), csr, src, o)
                assert np.array_equal(d, want), (w, len(ci), o is None)
    assert copy is not None
