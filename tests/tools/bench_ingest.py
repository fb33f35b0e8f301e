"""Ingest timing: write an R-MAT .mtx, load it with the engine's loader
(parallel line parser + parallel stable COO->CSR) and with the reference's own loader
(oracle/_ref/libgunrock_ref_cpu.so, compiled from /root/reference sources) when present.
    python tests/tools/bench_ingest.py [entries, default 8000000] [symmetric 0|1]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gunrock_amd as gr
import oracle_lib as O

entries = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
sym = len(sys.argv) > 2 and sys.argv[2] == "1"
V = max(1024, entries // 14)
_, c = gr.generate("rmat", V, entries, seed=11)
rows = np.repeat(np.arange(V, dtype=np.int64), np.diff(c.row_offsets)) + 1
cols = c.column_indices.astype(np.int64) + 1
if sym:
    rows, cols = np.maximum(rows, cols), np.minimum(rows, cols)
path = os.path.join(tempfile.mkdtemp(), "ingest.mtx")
t0 = time.time()
with open(path, "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate pattern %s\n%d %d %d\n" % ("symmetric" if sym else "general", V, V, entries))
    np.savetxt(f, np.stack([rows, cols], 1), fmt="%d %d")
size = os.path.getsize(path)
print("wrote %s: %d entries, %.1f MB (%.1f s)" % (path, entries, size / 1e6, time.time() - t0))
res = {}
for threads in (1, 0):
    if threads:
        os.environ["GRX_HOST_THREADS"] = str(threads)
    else:
        os.environ.pop("GRX_HOST_THREADS", None)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        props, coo = gr.matrix_market_t().load(path)
        best = min(best, time.perf_counter() - t0)
    res["ours_%s" % (threads or "all")] = best
    print("engine loader, threads=%s: %.3f s  (%.0f MB/s, %.1f M entries/s)" % (threads or os.cpu_count(), best, size / 1e6 / best, entries / 1e6 / best))
want = None
if O.have_ref_cpu():
    t0 = time.perf_counter()
    ref = O.ref_load_mtx(path)
    t_ref = time.perf_counter() - t0
    print("reference loader (fscanf per entry + from_coo): %.3f s -> speed-up %.1fx" % (t_ref, t_ref / res["ours_all"]))
    csr = gr.csr_t().from_coo(coo)
    print("identical CSR:", bool(np.array_equal(csr.row_offsets, ref.row_offsets) and np.array_equal(csr.column_indices, ref.column_indices)))
os.remove(path)
