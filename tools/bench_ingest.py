"""CPU-only: MatrixMarket ingest of the product's loader against the reference's own loader compiled here
(oracle/_ref, io/matrix_market.hxx:99-254 + formats/csr.hxx:81-140), on a generated pattern file:
    python tools/bench_ingest.py [edges, default 20000000]
Both produce a host CSR; the arrays are compared element by element."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402  (maps oracle/_ref before torch comes in)

have_ref = O.have_ref_cpu()
R = O.ref_cpu() if have_ref else None
import gunrock_amd as gr  # noqa: E402
from gunrock_amd import _capi  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
V = 1 << 21
rng = np.random.default_rng(1)
rows = rng.integers(1, V + 1, size=E, dtype=np.int64)
cols = rng.integers(1, V + 1, size=E, dtype=np.int64)
path = "/tmp/ingest_%d.mtx" % E
t0 = time.perf_counter()
with open(path, "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate pattern general\n%d %d %d\n" % (V, V, E))
    np.savetxt(f, np.stack([rows, cols], axis=1), fmt="%d")
print("wrote %s: %.1f MB in %.1f s" % (path, os.path.getsize(path) / 1e6, time.perf_counter() - t0), flush=True)

L = _capi.lib()
best = None
for _ in range(3):
    h = C.c_void_p()
    t0 = time.perf_counter()
    _capi.check(L.grx_host_csr_load_mtx(path.encode(), C.byref(h)))
    t = time.perf_counter() - t0
    best = t if best is None else min(best, t)
    if _ < 2:
        L.grx_host_csr_destroy(h)
ro, ci, x, props = gr._host_csr_to_numpy(h)
L.grx_host_csr_destroy(h)
print("ours (grx_host_csr_load_mtx, %d host threads): %.2f s best of 3 = %.1f M entries/s, %.0f MB/s of text"
      % (os.cpu_count(), best, E / best / 1e6, os.path.getsize(path) / best / 1e6), flush=True)
if have_ref:
    Vr, Er, pr = C.c_int(), C.c_int(), C.c_int()
    pro, pci, pw = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()
    t0 = time.perf_counter()
    rc = R.ref_load_mtx(path.encode(), C.byref(Vr), C.byref(Er), C.byref(pro), C.byref(pci), C.byref(pw), C.byref(pr))
    tr = time.perf_counter() - t0
    assert rc == 0, rc
    rro = np.ctypeslib.as_array(pro, shape=(Vr.value + 1,))
    rci = np.ctypeslib.as_array(pci, shape=(Er.value,))
    same = bool(np.array_equal(rro, ro) and np.array_equal(rci, ci))
    print("reference (matrix_market_t::load + csr_t::from_coo, one thread): %.2f s = %.1f M entries/s; CSR byte-equal to ours: %s; ours is %.1fx"
          % (tr, E / tr / 1e6, same, tr / best), flush=True)
else:
    print("oracle/_ref not built here: reference leg skipped")
os.remove(path)
