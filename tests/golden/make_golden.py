"""Generate the golden fixtures in tests/golden/ from the REFERENCE itself.

Run in the build container (needs /root/reference and oracle/_ref built with
`make -C oracle ref_cpu`).  Outputs are committed; the GPU box never sees
/root/reference.

  chesapeake.mtx        copy of the reference's dataset (datasets/chesapeake/),
                        the graph of its README example and SURVEY 8c golden vector
  tiny_real_general.mtx hand-written general real matrix (loader edge cases)
  golden.npz            CSR arrays produced by the reference's own loader +
                        from_coo, and distances produced by the reference's own
                        CPU oracle (bfs_cpu.hxx / sssp_cpu.hxx), for several
                        graphs and sources.
"""
import hashlib
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402

REF = "/root/reference"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert O.have_ref_cpu(), "build oracle/_ref first: make -C oracle ref_cpu"
    out = {}
    # 1. chesapeake: loader + BFS + SSSP from the reference
    src_mtx = os.path.join(REF, "datasets/chesapeake/chesapeake.mtx")
    dst_mtx = os.path.join(HERE, "chesapeake.mtx")
    shutil.copyfile(src_mtx, dst_mtx)
    g = O.ref_load_mtx(dst_mtx)
    out["chesapeake_ro"] = g.row_offsets
    out["chesapeake_ci"] = g.column_indices
    out["chesapeake_w"] = g.values
    out["chesapeake_props"] = np.array([g.props["directed"], g.props["weighted"], g.props["symmetric"]], dtype=np.int32)
    for s in (0, 5, 38):
        out["chesapeake_bfs_%d" % s] = O.ref_bfs_cpu(g, s)[0]
        out["chesapeake_sssp_%d" % s] = O.ref_sssp_cpu(g, s)[0]

    # 2. bips98_606 (real general, 7135 V / 34738 E): loader parity by hash
    b = O.ref_load_mtx(os.path.join(REF, "datasets/bips98_606/bips98_606.mtx"))
    out["bips_shape"] = np.array([b.n_vertices, b.n_edges], dtype=np.int64)
    out["bips_sha"] = np.array([sha(b.row_offsets), sha(b.column_indices), sha(b.values)])

    # 3. a small general real matrix with duplicates, a self loop, an isolated
    #    vertex and unsorted rows: loader + from_coo order
    tiny = os.path.join(HERE, "tiny_real_general.mtx")
    with open(tiny, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% comment line\n%\n6 6 9\n"
                "3 1 2.5\n1 2 1.0\n1 2 4.0\n2 2 7.0\n5 1 0.5\n1 6 3.0\n3 4 1.5\n6 5 2.0\n1 3 9.0\n")
    t = O.ref_load_mtx(tiny)
    out["tiny_ro"], out["tiny_ci"], out["tiny_w"] = t.row_offsets, t.column_indices, t.values
    out["tiny_sssp_0"] = O.ref_sssp_cpu(t, 0)[0]
    out["tiny_bfs_0"] = O.ref_bfs_cpu(t, 0)[0]
    # symmetric integer variant
    tsym = os.path.join(HERE, "tiny_int_symmetric.mtx")
    with open(tsym, "w") as f:
        f.write("%%MatrixMarket matrix coordinate integer symmetric\n5 5 6\n"
                "2 1 3\n3 1 10\n3 2 4\n4 3 2\n5 5 8\n5 4 1\n")
    ts = O.ref_load_mtx(tsym)
    out["tsym_ro"], out["tsym_ci"], out["tsym_w"] = ts.row_offsets, ts.column_indices, ts.values
    out["tsym_props"] = np.array([ts.props["directed"], ts.props["weighted"], ts.props["symmetric"]], dtype=np.int32)
    out["tsym_sssp_0"] = O.ref_sssp_cpu(ts, 0)[0]

    # 4. seeded synthetic graphs (arrays stored so the fixture does not depend
    #    on the generator): R-MAT directed, weighted lattice
    import gunrock_amd as gr
    _, c = gr.generate("rmat", 2000, 16000, seed=7)
    g2 = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    src = int(np.argmax(np.diff(g2.row_offsets)))
    out["rmat_ro"], out["rmat_ci"] = g2.row_offsets, g2.column_indices
    out["rmat_src"] = np.array([src], dtype=np.int32)
    out["rmat_bfs"] = O.ref_bfs_cpu(g2, src)[0]
    _, c = gr.generate("road", 40 * 40, a=0.7, c=1.0, seed=11)
    g3 = O.Csr(c.row_offsets, c.column_indices, c.nonzero_values)
    out["road_ro"], out["road_ci"], out["road_w"] = g3.row_offsets, g3.column_indices, g3.values
    rsrc = next(v for v in range(g3.n_vertices)
                if (O.bfs_queue(g3, v)[0] != np.iinfo(np.int32).max).sum() > g3.n_vertices // 2)
    out["road_src"] = np.array([rsrc], dtype=np.int32)
    out["road_sssp"] = O.ref_sssp_cpu(g3, rsrc)[0]
    out["road_bfs"] = O.ref_bfs_cpu(g3, rsrc)[0]

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    print("wrote golden.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
