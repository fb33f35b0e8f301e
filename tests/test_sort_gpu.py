"""The stable LSD radix sort behind the per-graph preprocessing (gunrock_amd/csrc/grx_sort.hpp: transpose, XCD-blocked
PageRank layout) against numpy's stable argsort: ragged sizes around the 64-lane step and the 4096-element tile, every key
width, heavy duplicates (stability is what makes the transpose reproducible), an optional second value."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sort(gr, ctx, keys, vals, vals2, bits):
    import torch
    from gunrock_amd import _capi
    k = torch.from_numpy(keys.view(np.int32)).cuda()
    v = torch.from_numpy(vals.view(np.int32)).cuda()
    v2 = torch.from_numpy(vals2.view(np.int32)).cuda() if vals2 is not None else None
    torch.cuda.synchronize()
    _capi.check(_capi.lib().grx_debug_radix_sort(ctx._h, C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()),
                                                 C.c_void_p(v2.data_ptr()) if v2 is not None else None, len(keys), bits))
    out = [k.cpu().numpy().view(np.uint32), v.cpu().numpy().view(np.uint32)]
    if v2 is not None:
        out.append(v2.cpu().numpy().view(np.uint32))
    return out


def test_stable_radix_sort_matches_numpy(gr, gpu_ctx):
    rng = np.random.default_rng(1)
    for n in (1, 2, 63, 64, 65, 1000, 4095, 4096, 4097, 8193, 100_003, 3_000_000):
        for bits in (1, 5, 9, 10, 17, 18, 23, 27, 31, 32):
            hi = (1 << bits) - 1
            keys = rng.integers(0, min(hi, 2 ** 32 - 1), n, dtype=np.uint64, endpoint=True).astype(np.uint32)
            if n > 1000:  # heavy duplicates: a few hot keys
                hot = rng.integers(0, min(hi, 2 ** 32 - 1), 3, dtype=np.uint64, endpoint=True).astype(np.uint32)
                sel = rng.random(n) < 0.5
                keys[sel] = hot[rng.integers(0, 3, int(sel.sum()))]
            vals = np.arange(n, dtype=np.uint32)
            vals2 = rng.integers(0, 2 ** 32 - 1, n, dtype=np.uint64).astype(np.uint32) if n % 2 else None
            order = np.argsort(keys, kind="stable")
            got = _sort(gr, gpu_ctx, keys, vals, vals2, bits)
            assert np.array_equal(got[0], keys[order]), (n, bits)
            assert np.array_equal(got[1], vals[order]), (n, bits)  # == order itself: stability
            if vals2 is not None:
                assert np.array_equal(got[2], vals2[order]), (n, bits)


def test_transpose_and_pagerank_are_reproducible_across_handles(gr, gpu_ctx):
    """Two FRESH graph handles over the same arrays: PageRank (fp32 sums over the in-edge lists, whose order the sort fixes)
    must agree bit for bit, on the plain and on the XCD-blocked layout, and so must direction-optimising BFS depths."""
    import torch
    _, c = gr.generate("rmat", 1 << 18, 5_000_000, seed=23)
    rng = np.random.default_rng(4)
    c.nonzero_values = (rng.random(c.number_of_nonzeros, dtype=np.float32) + np.float32(0.25)).astype(np.float32)
    props = gr.graph_properties_t(True, True, False)
    ranks, depths = [], []
    src = int(np.argmax(np.diff(c.row_offsets)))
    for _ in range(3):
        G = gr.build_graph(props, c, gpu_ctx)
        for flags in (0x40, 0x80):  # never / always the XCD-blocked layout
            p = torch.zeros(c.number_of_rows, dtype=torch.float32, device="cuda:0")
            res = gr.pr_result_t(p)
            gr.pr_run(G, gr.pr_param_t(0.85, 1e-6, gr.options_t(engine_flags=flags)), res, gpu_ctx)
            ranks.append((flags, p.cpu().numpy().copy(), res.iterations))
        d = torch.empty(c.number_of_rows, dtype=torch.int32, device="cuda:0")
        gr.bfs(G, src, d, None, gpu_ctx, gr.options_t(advance_direction=gr.optimized))
        depths.append(d.cpu().numpy().copy())
        del G
    for flags in (0x40, 0x80):
        same = [r for r in ranks if r[0] == flags]
        for other in same[1:]:
            assert other[2] == same[0][2] and np.array_equal(other[1], same[0][1]), flags
    assert all(np.array_equal(depths[0], x) for x in depths[1:])


def test_device_coo_to_csr_is_byte_identical_to_the_host_builder(gr, gpu_ctx):
    """grx_csr_from_coo_device against grx_host_csr_from_coo (the reference's stable row bucket sort, formats/csr.hxx:81-140):
    unsorted triples with duplicates and self loops, empty rows, a pattern (no values), sizes around the sort's tile, nnz = 0;
    a row index outside [0, n_rows) is refused."""
    import torch
    rng = np.random.default_rng(7)
    for n_rows, nnz in ((1, 1), (5, 0), (7, 40), (1000, 4096), (1000, 4097), (300_000, 2_000_003), (17, 100_000)):
        I = rng.integers(0, n_rows, nnz).astype(np.int32)
        J = rng.integers(0, n_rows, nnz).astype(np.int32)
        if nnz > 10:
            I[:5] = I[5:10]          # duplicates
            J[:5] = J[5:10]
            J[10:13] = I[10:13]      # self loops
            I[I == (n_rows // 2)] = 0  # an empty row
        X = (rng.random(nnz, dtype=np.float32) * 9 + 1).astype(np.float32)
        coo = gr.coo_t()
        coo.number_of_rows = coo.number_of_columns = n_rows
        coo.number_of_nonzeros = nnz
        coo.row_indices, coo.column_indices, coo.nonzero_values = I, J, X
        want = gr.csr_t().from_coo(coo)
        for vals in (X, None):
            ro, ci, x = gr.csr_t.from_coo_device(torch.from_numpy(I).cuda(), torch.from_numpy(J).cuda(),
                                                 None if vals is None else torch.from_numpy(vals).cuda(), n_rows, gpu_ctx)
            assert np.array_equal(ro.cpu().numpy(), want.row_offsets), (n_rows, nnz)
            assert np.array_equal(ci.cpu().numpy(), want.column_indices), (n_rows, nnz)
            if vals is not None:
                assert np.array_equal(x.cpu().numpy(), want.nonzero_values), (n_rows, nnz)
    bad = torch.tensor([0, 3, 9], dtype=torch.int32, device="cuda")
    with pytest.raises(gr.GrxError):
        gr.csr_t.from_coo_device(bad, bad.clone(), None, 5, gpu_ctx)
