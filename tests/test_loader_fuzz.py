"""Differential test of the MatrixMarket ingest on randomised small files: the product's loader
(grx_host_csr_load_mtx, multi-threaded) against the REFERENCE's own loader compiled here (oracle/_ref: io/matrix_market.hxx:99-254
+ formats/csr.hxx:81-140) and against the oracle's C restatement -- properties, offsets, column order and values must be identical.
CPU only; the reference leg runs where oracle/_ref exists (this container), the restatement leg everywhere."""
import os

import numpy as np
import pytest

import oracle_lib as O


def write_mtx(path, rng, field, scheme, rows, cols, nnz, comments, crlf):
    eol = "\r\n" if crlf else "\n"
    lines = ["%%MatrixMarket matrix coordinate " + field + " " + scheme]
    for k in range(comments):
        lines.append("% comment line " + str(k) + "  with trailing blanks   ")
    if scheme == "general":
        I = rng.integers(1, rows + 1, size=nnz)
        J = rng.integers(1, cols + 1, size=nnz)
    else:  # lower triangle (with diagonal for symmetric, strictly below for skew-symmetric)
        I = rng.integers(1, rows + 1, size=nnz)
        J = np.array([rng.integers(1, i + 1) if scheme == "symmetric" else rng.integers(1, max(i, 2)) for i in I])
        if scheme == "skew-symmetric":
            keep = J < I
            I, J = I[keep], J[keep]
    n = len(I)
    lines.append("%d %d %d" % (rows, cols, n))
    for k in range(n):
        if field == "pattern":
            lines.append("%d %d" % (I[k], J[k]))
        elif field == "integer":
            lines.append("%d %d %d" % (I[k], J[k], int(rng.integers(-50, 50))))
        else:
            lines.append("%d  %d\t%.6g" % (I[k], J[k], float(rng.normal()) * 10.0))
    with open(path, "w", newline="") as f:
        f.write(eol.join(lines) + eol)


CASES = [(field, scheme) for field in ("pattern", "real", "integer") for scheme in ("general", "symmetric", "skew-symmetric")
         if not (field == "pattern" and scheme == "skew-symmetric")]


@pytest.mark.parametrize("seed", range(4))
def test_loader_matches_reference_and_restatement_on_random_files(gr, tmp_path, seed):
    rng = np.random.default_rng(100 + seed)
    have_ref = O.have_ref_cpu()
    for field, scheme in CASES:
        n = int(rng.integers(1, 60))
        rows = cols = n  # the reference's loader is for graphs: square
        nnz = int(rng.integers(0, 6 * n + 1))
        p = str(tmp_path / ("f_%s_%s_%d.mtx" % (field, scheme, seed)))
        write_mtx(p, rng, field, scheme, rows, cols, nnz, comments=int(rng.integers(0, 4)), crlf=bool(seed & 1))
        props, coo = gr.matrix_market_t().load(p)
        csr = gr.csr_t().from_coo(coo)
        legs = [("restatement", O.load_mtx(p))]
        if have_ref:
            legs.append(("reference", O.ref_load_mtx(p)))
        for name, want in legs:
            ctx = (name, field, scheme, seed, n, nnz)
            assert np.array_equal(csr.row_offsets, want.row_offsets), ctx
            assert np.array_equal(csr.column_indices, want.column_indices), ctx
            assert np.array_equal(csr.nonzero_values, want.values), ctx
            wp = want.props if name == "reference" else None  # (the restatement's helper does not carry them)
            if wp:
                assert (bool(props.directed), bool(props.weighted), bool(props.symmetric)) == \
                       (bool(wp["directed"]), bool(wp["weighted"]), bool(wp["symmetric"])), ctx
