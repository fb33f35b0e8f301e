"""Generate tests/golden/golden_pr.npz: PageRank vectors + iteration counts produced by the REFERENCE's own GPU
path (oracle/_ref/libgunrock_ref_gpu.so = the reference's pr.hxx compiled from /root/reference by oracle/Makefile).

The reference has no CPU PageRank, no PR test and no --validate for it, so its GPU path is the only
thing that can pin the PR half of the oracle.  It needs a GPU: run on the GPU box,

    gpurun -- 'python tests/golden/make_golden_pr.py'      # writes gpurun_out/golden_pr.npz

then copy gpurun_out/golden_pr.npz to tests/golden/golden_pr.npz and commit it.  The graphs are the ones already
in golden.npz (reference loader output / seeded arrays), where the reference's atomicAdd-order noise is < 1e-7;
each is run REPEATS times and the spread between the runs is stored beside the vector, so a test knows how much of
a difference is the reference's own nondeterminism.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle_lib as O  # noqa: E402

REPEATS = 5


def cases(golden):
    ones = lambda k: np.ones(len(golden[k]), np.float32)
    yield "chesapeake", O.Csr(golden["chesapeake_ro"], golden["chesapeake_ci"], golden["chesapeake_w"]), 0.85, 1e-6
    yield "rmat", O.Csr(golden["rmat_ro"], golden["rmat_ci"], ones("rmat_ci")), 0.85, 1e-6
    yield "rmat_a50", O.Csr(golden["rmat_ro"], golden["rmat_ci"], ones("rmat_ci")), 0.5, 1e-6
    yield "rmat_tol8", O.Csr(golden["rmat_ro"], golden["rmat_ci"], ones("rmat_ci")), 0.85, 1e-8
    yield "road", O.Csr(golden["road_ro"], golden["road_ci"], golden["road_w"]), 0.85, 1e-6
    yield "tiny", O.Csr(golden["tiny_ro"], golden["tiny_ci"], golden["tiny_w"]), 0.85, 1e-7
    yield "tsym", O.Csr(golden["tsym_ro"], golden["tsym_ci"], golden["tsym_w"]), 0.85, 1e-6


def main():
    assert O.have_ref_gpu(), "oracle/_ref/libgunrock_ref_gpu.so missing: make -C oracle ref_gpu (build container)"
    golden = np.load(os.path.join(HERE, "golden.npz"))
    out = {}
    for name, g, alpha, tol in cases(golden):
        with O.RefGpuGraph(g) as R:
            runs = [R.pr(alpha, tol) for _ in range(REPEATS)]
            its = sorted(set(it for _, it, _ in runs))
            p0, it0, _ = runs[0]
            spread = max(float(np.abs(p0.astype(np.float64) - p).max()) for p, _, _ in runs)
            # every iterate of the reference up to its own stop, for the per-iteration comparison
            iterates = np.stack([R.pr(alpha, tol, force_iterations=k)[0] for k in range(1, it0 + 1)])
        out[name + "_p"] = p0
        out[name + "_iterations"] = np.array(its, np.int32)
        out[name + "_spread"] = np.array([spread])
        out[name + "_param"] = np.array([alpha, tol])
        out[name + "_iterates"] = iterates
        print(name, "V", g.n_vertices, "E", g.n_edges, "iterations", its, "run-to-run spread %.3g" % spread)
    dst = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dst, exist_ok=True)
    np.savez_compressed(os.path.join(dst, "golden_pr.npz"), **out)
    print("wrote gpurun_out/golden_pr.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
