"""ctypes loader for the test-only CPU oracle (oracle/liboracle.so) and, when it
has been built, the reference's own code (oracle/_ref/*.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = os.path.join(ORACLE_DIR, "liboracle.so")

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class _Coo(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("nnz", C.c_int32),
                ("row_indices", C.POINTER(C.c_int32)),
                ("column_indices", C.POINTER(C.c_int32)),
                ("nonzero_values", C.POINTER(C.c_float)),
                ("directed", C.c_int32), ("weighted", C.c_int32),
                ("symmetric", C.c_int32)]


class _Csr(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("nnz", C.c_int32),
                ("row_offsets", C.POINTER(C.c_int32)),
                ("column_indices", C.POINTER(C.c_int32)),
                ("nonzero_values", C.POINTER(C.c_float))]


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle.c", "oracle_omp.c", "oracle.h")]
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"],
                              stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(_LIB)
        L.orc_mtx_load.argtypes = [C.c_char_p, C.POINTER(_Coo)]
        L.orc_mtx_load.restype = C.c_int
        L.orc_csr_from_coo.argtypes = [C.POINTER(_Coo), C.POINTER(_Csr)]
        L.orc_csr_from_coo.restype = C.c_int
        L.orc_coo_free.argtypes = [C.POINTER(_Coo)]
        L.orc_csr_free.argtypes = [C.POINTER(_Csr)]
        L.orc_bfs.argtypes = [C.c_int32, i32p, i32p, C.c_int32, i32p]
        L.orc_bfs.restype = C.c_double
        L.orc_bfs_queue.argtypes = [C.c_int32, i32p, i32p, C.c_int32, i32p,
                                    C.POINTER(C.c_int64)]
        L.orc_bfs_queue.restype = C.c_double
        L.orc_sssp.argtypes = [C.c_int32, i32p, i32p, f32p, C.c_int32, f32p]
        L.orc_sssp.restype = C.c_double
        L.orc_sssp_budget.argtypes = [C.c_int32, i32p, i32p, f32p, C.c_int32, f32p, C.c_double,
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.orc_sssp_budget.restype = C.c_double
        L.orc_pr_f32.argtypes = [C.c_int32, i32p, i32p, f32p, C.c_float,
                                 C.c_float, C.c_int, f32p, C.POINTER(C.c_double)]
        L.orc_pr_f32.restype = C.c_int
        L.orc_pr_f64.argtypes = [C.c_int32, i32p, i32p, f32p, C.c_double,
                                 C.c_double, C.c_int, C.c_int, f64p,
                                 C.POINTER(C.c_double)]
        L.orc_pr_f64.restype = C.c_int
        L.orc_check_bfs.argtypes = [C.c_int32, i32p, i32p, C.c_int32, i32p]
        L.orc_check_bfs.restype = C.c_int64
        L.orc_check_sssp.argtypes = [C.c_int32, i32p, i32p, f32p, C.c_int32, f32p]
        L.orc_check_sssp.restype = C.c_int64
        L.orc_omp_threads.restype = C.c_int
        L.orc_bfs_omp.argtypes = [C.c_int32, i32p, i32p, C.c_int32, i32p, C.POINTER(C.c_int64)]
        L.orc_bfs_omp.restype = C.c_double
        L.orc_sssp_omp.argtypes = [C.c_int32, i32p, i32p, C.c_void_p, C.c_int32, f32p, C.c_double,
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_sssp_omp.restype = C.c_double
        L.orc_pr_omp.argtypes = [C.c_int32, i32p, i32p, C.c_void_p, C.c_float, C.c_int, f32p]
        L.orc_pr_omp.restype = C.c_double
        L.orc_pr_f64_trace.argtypes = [C.c_int32, i32p, i32p, C.c_void_p, C.c_double, C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p), f64p, f64p, C.c_void_p]
        L.orc_pr_f64_trace.restype = C.c_int
        _lib = L
    return _lib


class Csr:
    """Host CSR as numpy arrays (int32 offsets/indices, float32 values)."""

    def __init__(self, row_offsets, column_indices, values, props=None):
        self.row_offsets = np.ascontiguousarray(row_offsets, dtype=np.int32)
        self.column_indices = np.ascontiguousarray(column_indices, dtype=np.int32)
        self.values = np.ascontiguousarray(values, dtype=np.float32)
        self.n_vertices = len(self.row_offsets) - 1
        self.n_edges = len(self.column_indices)
        self.props = props or {"directed": 1, "weighted": 1, "symmetric": 0}


def load_mtx(path):
    """(.mtx) -> Csr via the oracle's restated loader + from_coo."""
    L = lib()
    coo = _Coo()
    rc = L.orc_mtx_load(path.encode(), C.byref(coo))
    if rc != 0:
        raise RuntimeError("orc_mtx_load failed with code %d for %s" % (rc, path))
    csr = _Csr()
    rc = L.orc_csr_from_coo(C.byref(coo), C.byref(csr))
    if rc != 0:
        raise RuntimeError("orc_csr_from_coo failed: %d" % rc)
    V, E = csr.rows, csr.nnz
    ro = np.ctypeslib.as_array(csr.row_offsets, shape=(V + 1,)).copy()
    ci = (np.ctypeslib.as_array(csr.column_indices, shape=(max(E, 1),))[:E]).copy()
    w = (np.ctypeslib.as_array(csr.nonzero_values, shape=(max(E, 1),))[:E]).copy()
    props = {"directed": coo.directed, "weighted": coo.weighted,
             "symmetric": coo.symmetric}
    L.orc_csr_free(C.byref(csr))
    L.orc_coo_free(C.byref(coo))
    return Csr(ro, ci, w, props)


def bfs(g, src):
    d = np.empty(g.n_vertices, dtype=np.int32)
    ms = lib().orc_bfs(g.n_vertices, g.row_offsets, g.column_indices, int(src), d)
    return d, ms


def bfs_queue(g, src):
    d = np.empty(g.n_vertices, dtype=np.int32)
    ev = C.c_int64(0)
    ms = lib().orc_bfs_queue(g.n_vertices, g.row_offsets, g.column_indices,
                             int(src), d, C.byref(ev))
    return d, ms, ev.value


def sssp(g, src):
    d = np.empty(g.n_vertices, dtype=np.float32)
    ms = lib().orc_sssp(g.n_vertices, g.row_offsets, g.column_indices, g.values,
                        int(src), d)
    return d, ms


def sssp_budget(g, src, budget_ms):
    """orc_sssp stopped after budget_ms -> (distances, ms, edges scanned, finished)."""
    d = np.empty(g.n_vertices, dtype=np.float32)
    ev, fin = C.c_int64(0), C.c_int32(0)
    ms = lib().orc_sssp_budget(g.n_vertices, g.row_offsets, g.column_indices, g.values, int(src), d,
                               float(budget_ms), C.byref(ev), C.byref(fin))
    return d, ms, ev.value, bool(fin.value)


def pr_f32(g, alpha=0.85, tol=1e-6, max_iterations=0):
    p = np.empty(g.n_vertices, dtype=np.float32)
    ms = C.c_double(0)
    it = lib().orc_pr_f32(g.n_vertices, g.row_offsets, g.column_indices, g.values,
                          alpha, tol, max_iterations, p, C.byref(ms))
    return p, it, ms.value


def pr_f64(g, alpha=0.85, tol=1e-6, max_iterations=0, force_iterations=0):
    p = np.empty(g.n_vertices, dtype=np.float64)
    ms = C.c_double(0)
    it = lib().orc_pr_f64(g.n_vertices, g.row_offsets, g.column_indices, g.values,
                          alpha, tol, max_iterations, force_iterations, p,
                          C.byref(ms))
    return p, it, ms.value


def omp_threads():
    return lib().orc_omp_threads()


def bfs_omp(g, src):
    """N-core level-synchronous BFS -> (depths, ms, edges visited)."""
    d = np.empty(g.n_vertices, dtype=np.int32)
    ev = C.c_int64(0)
    ms = lib().orc_bfs_omp(g.n_vertices, g.row_offsets, g.column_indices, int(src), d, C.byref(ev))
    return d, ms, ev.value


def sssp_omp(g, src, budget_ms=0.0, unit_weights=False):
    """N-core frontier Bellman-Ford -> (distances, ms, edges relaxed, iterations, finished)."""
    d = np.empty(g.n_vertices, dtype=np.float32)
    ev, it, fin = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    w = None if unit_weights else g.values.ctypes.data_as(C.c_void_p)
    ms = lib().orc_sssp_omp(g.n_vertices, g.row_offsets, g.column_indices, w, int(src), d,
                            float(budget_ms), C.byref(ev), C.byref(it), C.byref(fin))
    return d, ms, ev.value, it.value, bool(fin.value)


def pr_omp(g, alpha=0.85, iterations=5, pattern=False):
    """N-core pull PageRank, exactly `iterations` iterations -> (p, ms)."""
    p = np.empty(g.n_vertices, dtype=np.float32)
    w = None if pattern else g.values.ctypes.data_as(C.c_void_p)
    ms = lib().orc_pr_omp(g.n_vertices, g.row_offsets, g.column_indices, w, alpha, int(iterations), p)
    return p, ms


def pr_f64_trace(g, n_iter, cmp=(), alpha=0.85, pattern=False, want_final=False):
    """float64 pull PageRank for n_iter iterations -> (delta[n_iter], err[len(cmp)][n_iter], p_final or None):
    delta[k-1] = max|p_k - p_{k-1}|, err[c][k-1] = max|cmp[c] - p_k| (cmp: float32 vectors)."""
    cmp = [np.ascontiguousarray(c, dtype=np.float32) for c in cmp]
    ptrs = (C.c_void_p * max(1, len(cmp)))(*[c.ctypes.data for c in cmp])
    delta = np.zeros(n_iter, dtype=np.float64)
    err = np.zeros(max(1, len(cmp)) * n_iter, dtype=np.float64)
    pf = np.empty(g.n_vertices, dtype=np.float64) if want_final else None
    w = None if pattern else g.values.ctypes.data_as(C.c_void_p)
    rc = lib().orc_pr_f64_trace(g.n_vertices, g.row_offsets, g.column_indices, w, alpha, int(n_iter), len(cmp),
                                ptrs, delta, err, pf.ctypes.data_as(C.c_void_p) if want_final else None)
    if rc != 0:
        raise MemoryError("orc_pr_f64_trace")
    return delta, err.reshape(max(1, len(cmp)), n_iter)[:len(cmp)], pf


def pr_iterations_from_trace(delta, tol=1e-6):
    """loop() executions of the float64 recurrence: the first k >= 1 with max|p_k - p_{k-1}| < tol
    (is_converged skips the check at iteration 0 and then compares the previous loop's norm: pr.hxx:172-195)."""
    for k, d in enumerate(delta, start=1):
        if d < tol:
            return k
    return None


def check_bfs(g, src, dist):
    return lib().orc_check_bfs(g.n_vertices, g.row_offsets, g.column_indices,
                               int(src), np.ascontiguousarray(dist, dtype=np.int32))


def check_sssp(g, src, dist):
    return lib().orc_check_sssp(g.n_vertices, g.row_offsets, g.column_indices,
                                g.values, int(src),
                                np.ascontiguousarray(dist, dtype=np.float32))


# --------------------------------------------------------------------------
# The reference's own code (oracle/_ref), when built.
# --------------------------------------------------------------------------
_REF_CPU = os.path.join(ORACLE_DIR, "_ref", "libgunrock_ref_cpu.so")
_REF_GPU = os.path.join(ORACLE_DIR, "_ref", "libgunrock_ref_gpu.so")
_ref_cpu = None
_ref_gpu = None


def _ref_loadable():
    """oracle/_ref/*.so are hipcc-built and carry HIP static initialisers.  Mapped lazily they share the HIP runtime
    already in the process (torch's, on a GPU box).  In a GPU-LESS process that has imported torch first, mapping
    them crashes (two runtimes, no device) -- there the reference cross-checks are skipped unless the library was
    mapped before torch came in (the default test order does that)."""
    import sys
    if _ref_cpu is not None or _ref_gpu is not None:
        return True
    t = sys.modules.get("torch")
    if t is None:
        return True
    try:
        return bool(t.cuda.is_available())
    except Exception:
        return False


def have_ref_cpu():
    return os.path.exists(_REF_CPU) and _ref_loadable()


def have_ref_gpu():
    return os.path.exists(_REF_GPU)


def _bind_cpu(L):
    L.ref_bfs_cpu.argtypes = [C.c_int, C.c_int, i32p, i32p, C.c_int, i32p]
    L.ref_bfs_cpu.restype = C.c_float
    L.ref_sssp_cpu.argtypes = [C.c_int, C.c_int, i32p, i32p, f32p, C.c_int, f32p]
    L.ref_sssp_cpu.restype = C.c_float
    L.ref_load_mtx.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.POINTER(C.POINTER(C.c_int)),
                               C.POINTER(C.POINTER(C.c_int)),
                               C.POINTER(C.POINTER(C.c_float)),
                               C.POINTER(C.c_int)]
    L.ref_load_mtx.restype = C.c_int
    L.ref_free.argtypes = [C.c_void_p]


def ref_cpu():
    global _ref_cpu
    if _ref_cpu is None:
        L = C.CDLL(_REF_CPU)
        _bind_cpu(L)
        _ref_cpu = L
    return _ref_cpu


def ref_gpu():
    global _ref_gpu
    if _ref_gpu is None:
        L = C.CDLL(_REF_GPU)
        _bind_cpu(L)
        L.ref_gpu_graph_create.argtypes = [C.c_int, C.c_int, i32p, i32p, f32p]
        L.ref_gpu_graph_create.restype = C.c_void_p
        L.ref_gpu_graph_destroy.argtypes = [C.c_void_p]
        L.ref_gpu_bfs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, i32p]
        L.ref_gpu_bfs.restype = C.c_float
        L.ref_gpu_sssp.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p]
        L.ref_gpu_sssp.restype = C.c_float
        L.ref_gpu_pr.argtypes = [C.c_void_p, C.c_float, C.c_float, f32p]
        L.ref_gpu_pr.restype = C.c_float
        L.ref_gpu_pr_iters.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, f32p, C.POINTER(C.c_int)]
        L.ref_gpu_pr_iters.restype = C.c_float
        _ref_gpu = L
    return _ref_gpu


def ref_bfs_cpu(g, src):
    d = np.empty(g.n_vertices, dtype=np.int32)
    ms = ref_cpu().ref_bfs_cpu(g.n_vertices, g.n_edges, g.row_offsets,
                               g.column_indices, int(src), d)
    return d, ms


def ref_sssp_cpu(g, src):
    d = np.empty(g.n_vertices, dtype=np.float32)
    ms = ref_cpu().ref_sssp_cpu(g.n_vertices, g.n_edges, g.row_offsets,
                                g.column_indices, g.values, int(src), d)
    return d, ms


def ref_load_mtx(path):
    L = ref_cpu()
    V, E = C.c_int(0), C.c_int(0)
    ro = C.POINTER(C.c_int)()
    ci = C.POINTER(C.c_int)()
    w = C.POINTER(C.c_float)()
    props = (C.c_int * 3)()
    L.ref_load_mtx(path.encode(), C.byref(V), C.byref(E), C.byref(ro), C.byref(ci),
                   C.byref(w), props)
    v, e = V.value, E.value
    a = np.ctypeslib.as_array(ro, shape=(v + 1,)).astype(np.int32).copy()
    b = np.ctypeslib.as_array(ci, shape=(max(e, 1),))[:e].astype(np.int32).copy()
    c = np.ctypeslib.as_array(w, shape=(max(e, 1),))[:e].astype(np.float32).copy()
    for ptr in (ro, ci, w):
        L.ref_free(C.cast(ptr, C.c_void_p))
    return Csr(a, b, c, {"directed": props[0], "weighted": props[1],
                         "symmetric": props[2]})


class RefGpuGraph:
    """The reference's own device CSR for graph g (oracle/_ref/libgunrock_ref_gpu.so); GPU box only."""

    def __init__(self, g):
        self.L = ref_gpu()
        self.g = g
        self.h = self.L.ref_gpu_graph_create(g.n_vertices, g.n_edges, g.row_offsets, g.column_indices, g.values)

    def pr(self, alpha=0.85, tol=1e-6, force_iterations=0):
        """The reference's GPU PageRank (body of pr::run, pr.hxx:211-236) -> (p, loop() executions, enact ms).
        force_iterations > 0: exactly that many iterations instead of the reference's convergence test."""
        p = np.empty(self.g.n_vertices, np.float32)
        it = C.c_int(0)
        ms = self.L.ref_gpu_pr_iters(self.h, alpha, tol, int(force_iterations), p, C.byref(it))
        if ms < 0:
            raise RuntimeError("the reference's GPU PageRank failed")
        return p, it.value, ms

    def close(self):
        if self.h:
            self.L.ref_gpu_graph_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
