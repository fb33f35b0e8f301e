"""Driver of tools/proto/jacobi_sublevels.c (CPU design study, no GPU): relaxations and passes of the level-synchronous
weighted SSSP on a bench stand-in when a fat level runs as K sequential sub-levels.
    python tools/proto/jacobi_sublevels.py [lj|kron] [K ...]
Model of the time, from profiles/history/r4_relax_kernel_trace_lj.txt (scatter + sweep of the binned levels: 3150 us for 306 M
relaxations = 10.3 us per million; head + no-op level kernel + launches: ~14 us per pass):  t = 10.3 us x Medges + 14 us x passes."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402  (host generator only)
from bench import WORKLOADS, pair_hash_weights  # noqa: E402

so = "/tmp/libjacobi_sublevels.so"
subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tools", "proto", "jacobi_sublevels.c")], check=True)
L = C.CDLL(so)
name = sys.argv[1] if len(sys.argv) > 1 else "lj"
Ks = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4, 8]
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
w = pair_hash_weights(csr)
ro = np.ascontiguousarray(csr.row_offsets, np.int32)
ci = np.ascontiguousarray(csr.column_indices, np.int32)
src = int(np.argmax(np.diff(ro)))
print("workload", name, "V", len(ro) - 1, "E", len(ci), "src", src, flush=True)
for mode, mname in ((0, "parts in vertex order"), (1, "parts by ascending label")):
    for K in Ks:
        if mode == 1 and K == 1:
            continue
        out = (C.c_longlong * 5)()
        per = (C.c_longlong * 64)()
        L.jacobi_sublevels(C.c_int(len(ro) - 1), ro.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                           w.ctypes.data_as(C.c_void_p), C.c_int(src), C.c_int(K), C.c_int(mode), out, per, C.c_int(64))
        levels, passes, edges, verts, bad = [int(x) for x in out]
        t = 10.3 * edges / 1e6 + 14.0 * passes
        print("K %d  %-26s levels %2d passes %3d relaxed %11d (%.2f per edge) mismatches %d | model %.2f ms"
              % (K, mname, levels, passes, edges, edges / len(ci), bad, t / 1e3), flush=True)
