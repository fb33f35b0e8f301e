"""Block-asynchronous relaxation for road-like graphs (gunrock_amd/csrc/grx_block.hip) WITHOUT a GPU: the library's host
emulation of the device schedule (grx_debug_block_search_host: the real partitioner and block structure, then the same
supersteps / buckets / local rounds / boundary pass the kernels run) must reach the oracle's depths and distances bit for
bit -- BFS (bfs_cpu.hxx:32-63) and weighted SSSP (sssp_cpu.hxx:36-67) -- for every block size and several bucket widths,
on lattices with many small components, on a sparse random graph and from isolated sources."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O


def _has_block():
    try:
        import torch  # noqa: F401  (load order: torch's HIP runtime before libgrx's -- the other way round no device is found)
        from gunrock_amd import _capi
        return bool(_capi.lib().grx_has_block_async())
    except Exception:  # noqa: BLE001
        return False


# (the default library does not carry this path since round 6: tests/test_block_variant.py runs this file against libgrx_block.so)
pytestmark = pytest.mark.skipif(not _has_block(), reason="library built without grx_block.hip (python -m gunrock_amd.build --with-block)")


class BlockStats(C.Structure):
    _fields_ = [("edges_relaxed", C.c_int64), ("activations", C.c_int64), ("cross_edges", C.c_int64),
                ("supersteps", C.c_int32), ("buckets", C.c_int32), ("blocks", C.c_int32), ("block_vertices", C.c_int32),
                ("build_ms", C.c_double)]


def _lib():
    from gunrock_amd import _capi
    L = _capi.lib()
    L.grx_debug_block_search_host.restype = C.c_int
    L.grx_debug_block_search_host.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p,
                                              C.POINTER(BlockStats)]
    return L, _capi


def _host_csr_from_arrays(L, capi, ro, ci, w):
    rows = np.repeat(np.arange(len(ro) - 1, dtype=np.int32), np.diff(ro)).astype(np.int32)
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    h = C.c_void_p()
    capi.check(L.grx_host_csr_from_coo(len(ro) - 1, len(ro) - 1, len(ci), rows.ctypes.data, ci.ctypes.data, w.ctypes.data,
                                       C.byref(h)))
    return h


def _search(L, capi, h, V, weighted, nv, src, delta):
    out = np.empty(V, dtype=np.uint32)
    st = BlockStats()
    bits = np.float32(delta).view(np.uint32) if weighted else np.uint32(delta)
    capi.check(L.grx_debug_block_search_host(h, int(weighted), nv, int(src), int(bits), out.ctypes.data, C.byref(st)))
    return out, st


@pytest.mark.parametrize("weighted", [False, True])
def test_host_emulation_matches_the_oracle_on_lattices(gr, weighted):
    L, capi = _lib()
    side = 300
    _, c = gr.generate("road", side * side, a=0.602, c=1.0 if weighted else 0.0, seed=7)
    g = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    h = _host_csr_from_arrays(L, capi, g.row_offsets, g.column_indices, g.values)
    deg = np.diff(g.row_offsets)
    sources = [(side // 2) * side + side // 2, 0, int(np.nonzero(deg == 0)[0][0])]
    try:
        for src in sources:
            if weighted:
                want = O.sssp(g, src)[0].view(np.uint32)
            else:
                want = O.bfs_queue(g, src)[0].view(np.uint32)
            for nv in ((2048, 4096) if weighted else (2048, 4096, 8192)):
                for delta in ((500.0, 20000.0, 1e9) if weighted else (1, 37, 256, 1 << 30)):
                    got, st = _search(L, capi, h, g.n_vertices, weighted, nv, src, delta)
                    assert np.array_equal(got, want), (weighted, src, nv, delta)
                    assert st.blocks * nv >= g.n_vertices and st.supersteps >= 1
    finally:
        L.grx_host_csr_destroy(h)


def test_host_emulation_sparse_random_graph_and_rejection(gr):
    L, capi = _lib()
    rng = np.random.default_rng(3)
    n, k = 60_000, 3
    ro = (np.arange(n + 1, dtype=np.int64) * k).astype(np.int32)
    ci = rng.integers(0, n, n * k).astype(np.int32)
    w = rng.integers(1, 64, n * k).astype(np.float32)
    g = O.Csr(ro, ci, w)
    h = _host_csr_from_arrays(L, capi, ro, ci, w)
    try:
        want = O.sssp(g, 5)[0].view(np.uint32)
        got, st = _search(L, capi, h, n, True, 2048, 5, 40.0)
        assert np.array_equal(got, want)
        assert st.cross_edges > n  # no locality: most edges leave their block -- slow, still exact
        gu = O.Csr(ro, ci, np.ones(len(ci), np.float32))
        got, _ = _search(L, capi, h, n, False, 4096, 5, 3)
        assert np.array_equal(got, O.bfs_queue(gu, 5)[0].view(np.uint32))
    finally:
        L.grx_host_csr_destroy(h)
    # a row that does not fit a block: the structure is refused (the engine then keeps its level-synchronous path)
    star_ro = np.concatenate([[0], np.full(20000, 19999)]).astype(np.int32)
    star_ci = np.arange(1, 20000, dtype=np.int32)
    h2 = _host_csr_from_arrays(L, capi, star_ro, star_ci, np.ones(len(star_ci), np.float32))
    try:
        out = np.empty(20000, dtype=np.uint32)
        rc = L.grx_debug_block_search_host(h2, 0, 2048, 0, 16, out.ctypes.data, None)
        assert rc != 0
    finally:
        L.grx_host_csr_destroy(h2)
