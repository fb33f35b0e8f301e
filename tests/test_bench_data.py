"""bench.py --data-dir / GRX_DATA_DIR: the published graphs replace the seeded stand-ins when their .mtx files are
present on the measurement box.  Host-side only: resolver + the engine's Matrix-Market loader."""
import os
import shutil

import numpy as np

import oracle_lib as O
from conftest import GOLDEN


def test_real_files_are_found_and_loaded_like_the_reference_loader(gr, tmp_path, golden):
    import bench
    ches = os.path.join(GOLDEN, "chesapeake.mtx")
    flat = tmp_path / "flat"
    nested = tmp_path / "nested"
    flat.mkdir()
    (nested / "road_usa").mkdir(parents=True)
    shutil.copy(ches, flat / "soc-LiveJournal1.mtx")
    shutil.copy(ches, nested / "road_usa" / "road_usa.mtx")
    assert bench.find_real("lj", str(flat)) == str(flat / "soc-LiveJournal1.mtx")
    assert bench.find_real("road", str(nested)) == str(nested / "road_usa" / "road_usa.mtx")
    assert bench.find_real("kron", str(flat)) is None and bench.find_real("lj", "") is None
    assert bench.find_real("small", str(flat)) is None  # no published counterpart
    props, csr, src, info = bench.load_workload(gr, "lj", str(flat))
    assert info["data"] == "real" and info["file"].endswith("soc-LiveJournal1.mtx")
    assert np.array_equal(csr.row_offsets, golden["chesapeake_ro"])      # the reference loader's own output
    assert np.array_equal(csr.column_indices, golden["chesapeake_ci"])
    assert np.array_equal(csr.nonzero_values, golden["chesapeake_w"])
    assert src == int(np.argmax(np.diff(golden["chesapeake_ro"])))
    # pattern file + weighted section: synthetic weights on the real topology, symmetric like the file
    props, csr, _, info = bench.load_workload(gr, "road", str(nested), weighted=True)
    assert info["data"].startswith("real topology") and props.weighted
    w = csr.nonzero_values
    assert w.min() >= 1 and w.max() <= 1000 and len(np.unique(w)) > 20
    rows = np.repeat(np.arange(csr.number_of_rows), np.diff(csr.row_offsets))
    fwd = {(int(u), int(v)): float(x) for u, v, x in zip(rows, csr.column_indices, w)}
    assert all(fwd[(v, u)] == x for (u, v), x in fwd.items())
    g = O.Csr(csr.row_offsets, csr.column_indices, w)
    assert O.check_sssp(g, 0, O.sssp(g, 0)[0]) == 0
    # no file: the seeded stand-in, unchanged
    props, csr, src, info = bench.load_workload(gr, "small", str(flat))
    assert info["data"] == "synthetic" and csr.number_of_nonzeros == 4_000_000
