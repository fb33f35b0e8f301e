"""The C-ABI library loads and exports every symbol include/grx.h declares; the
host-side ingest (loader, from_coo, binary CSR, generators) matches the
reference-produced goldens.  No GPU needed, no compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN, ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "grx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(grx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(gr):
    from gunrock_amd import _capi
    L = C.CDLL(_capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert b"gfx950" in _capi.lib().grx_version_string()


def test_options_default_matches_reference_defaults(gr):
    from gunrock_amd import _capi
    o = _capi.grx_options_t()
    _capi.lib().grx_options_default(C.byref(o))
    # include/gunrock/algorithms/algorithms.hxx:29-47
    assert o.advance_load_balance == gr.block_mapped == 2
    assert o.filter_algorithm == gr.predicated == 1
    assert (o.enable_filter, o.enable_uniquify, o.best_effort_uniquify) == (0, 0, 1)
    assert o.uniquify_percent == 100.0
    # enum integer values of operators/configs.hxx:52-112
    assert (gr.thread_mapped, gr.warp_mapped, gr.block_mapped, gr.bucketing, gr.merge_path,
            gr.merge_path_v2, gr.work_stealing) == (0, 1, 2, 3, 4, 5, 6)
    assert (gr.remove, gr.predicated, gr.compact, gr.bypass) == (0, 1, 2, 3)


def test_product_loader_matches_reference_loader(gr, golden):
    props, coo = gr.matrix_market_t().load(os.path.join(GOLDEN, "chesapeake.mtx"))
    csr = gr.csr_t().from_coo(coo)
    assert (props.directed, props.weighted, props.symmetric) == (False, False, True)
    assert np.array_equal(csr.row_offsets, golden["chesapeake_ro"])
    assert np.array_equal(csr.column_indices, golden["chesapeake_ci"])
    assert np.array_equal(csr.nonzero_values, golden["chesapeake_w"])
    for name, key in (("tiny_real_general.mtx", "tiny"), ("tiny_int_symmetric.mtx", "tsym")):
        p, c = gr.matrix_market_t().load(os.path.join(GOLDEN, name))
        k = gr.csr_t().from_coo(c)
        assert np.array_equal(k.row_offsets, golden[key + "_ro"])
        assert np.array_equal(k.column_indices, golden[key + "_ci"])
        assert np.array_equal(k.nonzero_values, golden[key + "_w"])


def test_from_coo_is_stable_and_keeps_duplicates(gr):
    coo = gr.coo_t(4, 4, 6)
    coo.row_indices[:] = [2, 0, 2, 0, 3, 2]
    coo.column_indices[:] = [1, 3, 1, 0, 3, 0]
    coo.nonzero_values[:] = [1, 2, 3, 4, 5, 6]
    csr = gr.csr_t().from_coo(coo)
    assert csr.row_offsets.tolist() == [0, 2, 2, 5, 6]
    assert csr.column_indices.tolist() == [3, 0, 1, 1, 0, 3]
    assert csr.nonzero_values.tolist() == [2, 4, 1, 3, 6, 5]


def test_loader_error_behaviour(gr, tmp_path):
    with pytest.raises(gr.GrxError, match="could not be opened"):
        gr.matrix_market_t().load(str(tmp_path / "nope.mtx"))
    p = tmp_path / "arr.mtx"
    p.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(gr.GrxError, match="not a sparse matrix"):
        gr.matrix_market_t().load(str(p))
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n2 2 1\n0 1\n")
    with pytest.raises(gr.GrxError, match="zero-indexed"):
        gr.matrix_market_t().load(str(p))
    p.write_text("%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1.0 0.0\n")
    with pytest.raises(gr.GrxError):
        gr.matrix_market_t().load(str(p))


def test_binary_csr_roundtrip(gr, golden, tmp_path):
    csr = gr.csr_t.from_arrays(golden["road_ro"], golden["road_ci"], golden["road_w"])
    f = str(tmp_path / "g.csr")
    csr.write_binary(f)
    # header {rows:i32, cols:i32, nnz:i32} then offsets, indices, values (formats/csr.hxx:142-228)
    hdr = np.fromfile(f, dtype=np.int32, count=3)
    assert hdr.tolist() == [csr.number_of_rows, csr.number_of_columns, csr.number_of_nonzeros]
    back = gr.csr_t().read_binary(f)
    assert np.array_equal(back.row_offsets, csr.row_offsets)
    assert np.array_equal(back.column_indices, csr.column_indices)
    assert np.array_equal(back.nonzero_values, csr.nonzero_values)


def test_generators_are_deterministic_and_well_formed(gr):
    p1, a = gr.generate("rmat", 5000, 40000, seed=42)
    p2, b = gr.generate("rmat", 5000, 40000, seed=42)
    assert np.array_equal(a.column_indices, b.column_indices) and np.array_equal(a.row_offsets, b.row_offsets)
    assert a.number_of_nonzeros == 40000 and p1.directed and not p1.weighted
    assert a.column_indices.min() >= 0 and a.column_indices.max() < 5000
    deg = np.diff(a.row_offsets)
    assert deg.max() > 20 * deg.mean()  # skewed
    ps, s = gr.generate("rmat_sym", 3000, 20000, seed=1)
    g = O.Csr(s.row_offsets, s.column_indices, s.nonzero_values)
    # symmetric: every (u,v) has (v,u)
    src = np.repeat(np.arange(3000), np.diff(g.row_offsets))
    fwd = set(zip(src.tolist(), g.column_indices.tolist()))
    assert all((v, u) in fwd for (u, v) in list(fwd)[:2000])
    pr, r = gr.generate("road", 30 * 30, a=0.7, c=1.0, seed=5)
    assert pr.symmetric and pr.weighted
    assert r.nonzero_values.min() >= 1 and r.nonzero_values.max() <= 1000
    assert np.all(r.nonzero_values == np.round(r.nonzero_values))


def _write_mtx(path, header, rows, cols, vals, sep="\n", extra_blank=False):
    with open(path, "w", newline="") as f:
        f.write(header + sep)
        f.write("% a comment" + sep)
        f.write("%d %d %d" % (rows.max() + 1 if len(rows) else 1, rows.max() + 1 if len(rows) else 1, len(rows)) + sep)
        for k in range(len(rows)):
            if extra_blank and k % 997 == 0:
                f.write("   " + sep)
            if vals is None:
                f.write("%d %d" % (rows[k] + 1, cols[k] + 1) + sep)
            else:
                f.write("%d\t%d  %s " % (rows[k] + 1, cols[k] + 1, vals[k]) + sep)


@pytest.mark.parametrize("kind", ["pattern general", "real symmetric", "integer general"])
def test_parallel_ingest_equals_oracle_loader(gr, tmp_path, kind, monkeypatch):
    """Files big enough for the multi-threaded line parser (>= 4096 entries), with the
    formatting real files have: CRLF, tabs, trailing blanks, blank lines, exponents,
    negative values, duplicates, self loops.  Checked against the oracle restatement of
    the reference loader (and the reference loader itself when oracle/_ref is built)."""
    rng = np.random.default_rng(7)
    n, nnz = 3000, 50000
    rows = rng.integers(0, n, nnz)
    cols = rng.integers(0, n, nnz)
    rows[-1] = n - 1  # make the declared size exact
    if kind == "real symmetric":
        lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
        rows, cols = hi, lo
        rows[-1] = n - 1
    vals = None
    if kind.startswith("real"):
        pool = ["1", "0.5", "-3.25", "1e-3", "2.5E+2", "123456789.125", "7.", ".5", "1e22", "3.0000000000000001e-5",
                "0.1234567890123456789", "-0"]
        vals = [pool[i] for i in rng.integers(0, len(pool), nnz)]
    elif kind.startswith("integer"):
        vals = [str(int(v)) for v in rng.integers(-50, 1000, nnz)]
    path = str(tmp_path / "big.mtx")
    for sep, blank in (("\n", False), ("\r\n", True)):
        _write_mtx(path, "%%MatrixMarket matrix coordinate " + kind, rows, cols, vals, sep=sep, extra_blank=blank)
        want = O.load_mtx(path)
        for threads in ("1", "5"):
            monkeypatch.setenv("GRX_HOST_THREADS", threads)
            props, coo = gr.matrix_market_t().load(path)
            csr = gr.csr_t().from_coo(coo)
            assert np.array_equal(csr.row_offsets, want.row_offsets)
            assert np.array_equal(csr.column_indices, want.column_indices)
            assert np.array_equal(csr.nonzero_values, want.values)  # bit-exact floats (strtod semantics)
        if O.have_ref_cpu():
            ref = O.ref_load_mtx(path)
            assert np.array_equal(ref.row_offsets, want.row_offsets) and np.array_equal(ref.values, want.values)


def test_parallel_ingest_falls_back_on_wrapped_entries(gr, tmp_path):
    """Entries wrapped over lines are legal for the reference's fscanf loop: the line
    parser must notice and hand over to the token parser."""
    rng = np.random.default_rng(3)
    n, nnz = 500, 6000
    rows, cols = rng.integers(0, n, nnz), rng.integers(0, n, nnz)
    rows[0] = n - 1
    path = str(tmp_path / "wrapped.mtx")
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate pattern general\n%d %d %d\n" % (n, n, nnz))
        for k in range(nnz):
            f.write("%d\n%d " % (rows[k] + 1, cols[k] + 1) if k % 2 else "%d %d\n" % (rows[k] + 1, cols[k] + 1))
    want = O.load_mtx(path)
    _, coo = gr.matrix_market_t().load(path)
    csr = gr.csr_t().from_coo(coo)
    assert np.array_equal(csr.row_offsets, want.row_offsets) and np.array_equal(csr.column_indices, want.column_indices)
    # an entry outside the declared matrix is an error, not an out-of-bounds write
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern general\n3 3 2\n1 2\n4 1\n")
    with pytest.raises(RuntimeError):
        gr.matrix_market_t().load(path)
