"""gunrock_amd -- Python host side of the MI355X-native frontier engine.

Mirrors the reference's Python module (python/src/gunrock/bindings.cu) name for
name: `matrix_market_t().load`, `csr_t().from_coo`, `build_graph`,
`multi_context_t`, `options_t`, `bfs`, `sssp`, `pr_run`, the `*_param_t` /
`*_result_t` holders and the operator enums, so that code written against
`import gunrock` runs with `import gunrock_amd as gunrock`.

All compute goes through the C ABI in include/grx.h (libgrx.so, hand-written
HIP for gfx950).  torch is used only for device memory and streams.  There is
no CPU fallback: importing works without a GPU (so the host logic is testable),
running an algorithm without the HIP library or a device raises.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (GrxError, grx_options_t, grx_run_stats_t, grx_level_profile_t, grx_block_stats_t,
                    thread_mapped, warp_mapped, block_mapped, bucketing, merge_path,
                    merge_path_v2, work_stealing, remove, predicated, compact, bypass,
                    unique, unique_copy, forward, backward, optimized,
                    FLAG_UNFUSED, FLAG_PROFILE, FLAG_SYNC_EACH_LEVEL, FLAG_ASYNC_RETURN, FLAG_LB_STRICT,
                    FLAG_SSSP_PLAIN, FLAG_SSSP_NEAR_FAR, FLAG_SSSP_NO_BFS, FLAG_SSSP_NO_BINS, FLAG_NO_BLOCK_ASYNC)

__all__ = ["memory_space_t", "graph_properties_t", "coo_t", "csr_t", "graph_t",
           "matrix_market_t", "build_graph", "multi_context_t", "options_t",
           "bfs", "sssp", "pr_run", "bfs_param_t", "sssp_param_t", "pr_param_t",
           "pr_result_t", "GrxError", "generate", "run_stats", "level_profile"]


class memory_space_t:  # include/gunrock/memory.hxx:33
    device = 0
    host = 1


class graph_properties_t:
    """include/gunrock/graph/properties.hxx:13-18 (same defaults)."""

    def __init__(self, directed=False, weighted=True, symmetric=True):
        self.directed = bool(directed)
        self.weighted = bool(weighted)
        self.symmetric = bool(symmetric)

    def __repr__(self):
        return "graph_properties_t(directed=%s, weighted=%s, symmetric=%s)" % (
            self.directed, self.weighted, self.symmetric)


class coo_t:
    """format::coo_t<host> (include/gunrock/formats/coo.hxx:23-46): host arrays."""

    def __init__(self, number_of_rows=0, number_of_columns=0, number_of_nonzeros=0):
        self.number_of_rows = int(number_of_rows)
        self.number_of_columns = int(number_of_columns)
        self.number_of_nonzeros = int(number_of_nonzeros)
        self.row_indices = np.zeros(self.number_of_nonzeros, dtype=np.int32)
        self.column_indices = np.zeros(self.number_of_nonzeros, dtype=np.int32)
        self.nonzero_values = np.zeros(self.number_of_nonzeros, dtype=np.float32)


def _host_csr_to_numpy(h):
    L = _capi.lib()
    V, E = C.c_int32(), C.c_int32()
    d, w, s = C.c_int32(), C.c_int32(), C.c_int32()
    _capi.check(L.grx_host_csr_info(h, C.byref(V), C.byref(E), C.byref(d), C.byref(w), C.byref(s)))
    v, e = V.value, E.value
    ro = np.ctypeslib.as_array(L.grx_host_csr_row_offsets(h), shape=(v + 1,)).copy()
    if e > 0:
        ci = np.ctypeslib.as_array(L.grx_host_csr_column_indices(h), shape=(e,)).copy()
        x = np.ctypeslib.as_array(L.grx_host_csr_values(h), shape=(e,)).copy()
    else:
        ci = np.zeros(0, dtype=np.int32)
        x = np.zeros(0, dtype=np.float32)
    props = graph_properties_t(directed=bool(d.value), weighted=bool(w.value),
                               symmetric=bool(s.value))
    return ro, ci, x, props


class csr_t:
    """format::csr_t (include/gunrock/formats/csr.hxx:27-69).

    Host arrays live in numpy (`row_offsets`, `column_indices`,
    `nonzero_values`); `.to_device()` uploads them once as torch tensors, which
    is what `build_graph` hands to the engine as a non-owning view.
    """

    def __init__(self, number_of_rows=0, number_of_columns=0, number_of_nonzeros=0):
        self.number_of_rows = int(number_of_rows)
        self.number_of_columns = int(number_of_columns)
        self.number_of_nonzeros = int(number_of_nonzeros)
        self.row_offsets = np.zeros(self.number_of_rows + 1, dtype=np.int32)
        self.column_indices = np.zeros(self.number_of_nonzeros, dtype=np.int32)
        self.nonzero_values = np.zeros(self.number_of_nonzeros, dtype=np.float32)
        self._device = None

    def _set(self, ro, ci, x, cols=None):
        self.row_offsets = np.ascontiguousarray(ro, dtype=np.int32)
        self.column_indices = np.ascontiguousarray(ci, dtype=np.int32)
        self.nonzero_values = np.ascontiguousarray(x, dtype=np.float32)
        self.number_of_rows = len(self.row_offsets) - 1
        self.number_of_columns = self.number_of_rows if cols is None else int(cols)
        self.number_of_nonzeros = len(self.column_indices)
        self._device = None
        return self

    @classmethod
    def from_arrays(cls, row_offsets, column_indices, values=None):
        ci = np.asarray(column_indices)
        x = np.ones(len(ci), dtype=np.float32) if values is None else values
        return cls()._set(row_offsets, ci, x)

    def from_coo(self, coo):
        """csr_t::from_coo (formats/csr.hxx:81-140): stable row bucket sort."""
        L = _capi.lib()
        h = C.c_void_p()
        I = np.ascontiguousarray(coo.row_indices, dtype=np.int32)
        J = np.ascontiguousarray(coo.column_indices, dtype=np.int32)
        X = np.ascontiguousarray(coo.nonzero_values, dtype=np.float32)
        _capi.check(L.grx_host_csr_from_coo(
            int(coo.number_of_rows), int(coo.number_of_columns), int(coo.number_of_nonzeros),
            I.ctypes.data, J.ctypes.data, X.ctypes.data, C.byref(h)))
        try:
            ro, ci, x, _ = _host_csr_to_numpy(h)
        finally:
            L.grx_host_csr_destroy(h)
        return self._set(ro, ci, x, coo.number_of_columns)

    @staticmethod
    def from_coo_device(row_indices, column_indices, values, number_of_rows, context):
        """csr_t::from_coo (formats/csr.hxx:81-140) on the DEVICE: torch tensors in (int32 rows / columns, float32 values or
        None, on the context's device), torch tensors out (row_offsets, column_indices, values or None) -- the same stable
        order as the host builder (entries of a row keep their input order), byte-identical to `from_coo`."""
        import torch
        nnz = int(row_indices.numel())
        dev = row_indices.device
        ro = torch.empty(int(number_of_rows) + 1, dtype=torch.int32, device=dev)
        ci = torch.empty(nnz, dtype=torch.int32, device=dev)
        x = None if values is None else torch.empty(nnz, dtype=torch.float32, device=dev)
        assert row_indices.dtype == torch.int32 and column_indices.dtype == torch.int32 and row_indices.is_contiguous() \
            and column_indices.is_contiguous() and (values is None or (values.dtype == torch.float32 and values.is_contiguous()))
        torch.cuda.current_stream(dev).synchronize()  # the inputs may come from torch's stream; the library uses its own
        _capi.check(_capi.lib().grx_csr_from_coo_device(
            context._h, int(number_of_rows), nnz, row_indices.data_ptr() if nnz else None,
            column_indices.data_ptr() if nnz else None, None if values is None else values.data_ptr(),
            ro.data_ptr(), ci.data_ptr() if nnz else None, None if x is None else x.data_ptr()))
        return ro, ci, x

    def read_binary(self, filename):
        """csr_t::read_binary (formats/csr.hxx:142-192)."""
        L = _capi.lib()
        h = C.c_void_p()
        _capi.check(L.grx_host_csr_read_binary(str(filename).encode(), C.byref(h)))
        try:
            ro, ci, x, _ = _host_csr_to_numpy(h)
        finally:
            L.grx_host_csr_destroy(h)
        return self._set(ro, ci, x)

    def write_binary(self, filename):
        """csr_t::write_binary (formats/csr.hxx:194-228)."""
        with open(filename, "wb") as f:
            np.array([self.number_of_rows, self.number_of_columns, self.number_of_nonzeros],
                     dtype=np.int32).tofile(f)
            self.row_offsets.tofile(f)
            self.column_indices.tofile(f)
            self.nonzero_values.tofile(f)

    def to_device(self, device="cuda:0"):
        import torch
        if self._device is None or str(self._device[0].device) != str(torch.device(device)):
            self._device = (torch.from_numpy(self.row_offsets).to(device),
                            torch.from_numpy(self.column_indices).to(device),
                            torch.from_numpy(self.nonzero_values).to(device))
        return self._device


class matrix_market_t:
    """io::matrix_market_t::load (include/gunrock/io/matrix_market.hxx:99-254)."""

    def __init__(self):
        self.filename = ""

    def load(self, filename):
        """-> (graph_properties_t, coo_t); pattern => 1.0 weights, symmetric =>
        each off-diagonal entry followed by its mirror."""
        self.filename = str(filename)
        L = _capi.lib()
        h = C.c_void_p()
        _capi.check(L.grx_host_csr_load_mtx(self.filename.encode(), C.byref(h)))
        try:
            ro, ci, x, props = _host_csr_to_numpy(h)
        finally:
            L.grx_host_csr_destroy(h)
        # The engine's loader already bucketed by row with the reference's
        # stable order; expose it back as COO (row-major, same order csr gives).
        n = len(ro) - 1
        coo = coo_t(n, n, len(ci))
        coo.row_indices = np.repeat(np.arange(n, dtype=np.int32), np.diff(ro))
        coo.column_indices = ci
        coo.nonzero_values = x
        return props, coo


class multi_context_t:
    """gcuda::multi_context_t(device) (include/gunrock/cuda/context.hxx:146-189)."""

    def __init__(self, device_id=0, stream=None):
        L = _capi.lib()
        self._h = C.c_void_p()
        self.device_id = int(device_id)
        sp = None
        self._stream_handle = None
        if stream is not None:
            self._stream_handle = int(getattr(stream, "cuda_stream", stream))
            sp = C.c_void_p(self._stream_handle)
        _capi.check(L.grx_context_create(self.device_id, sp, C.byref(self._h)))

    def synchronize(self):
        _capi.check(_capi.lib().grx_context_synchronize(self._h))

    def size(self):
        return 1

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _capi.lib().grx_context_destroy(self._h)
                self._h = None
        except Exception:
            pass


class options_t:
    """gunrock::options_t (include/gunrock/algorithms/algorithms.hxx:27-72)."""

    def __init__(self, advance_load_balance=block_mapped, filter_algorithm=predicated,
                 enable_filter=False, enable_uniquify=False, uniquify_algorithm=unique,
                 best_effort_uniquify=True, uniquify_percent=100.0, engine_flags=0,
                 advance_direction=forward, max_iterations=0):
        self.advance_load_balance = advance_load_balance
        self.filter_algorithm = filter_algorithm
        self.enable_filter = enable_filter
        self.enable_uniquify = enable_uniquify
        self.uniquify_algorithm = uniquify_algorithm
        self.best_effort_uniquify = best_effort_uniquify
        self.uniquify_percent = uniquify_percent
        self.engine_flags = engine_flags
        self.advance_direction = advance_direction
        self.max_iterations = max_iterations

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        object.__setattr__(self, "_cached", None)

    def _c(self):
        o = self._cached
        if o is not None:
            return o  # (the engine reads the struct, it never writes it)
        o = grx_options_t()
        o.advance_load_balance = int(self.advance_load_balance)
        o.filter_algorithm = int(self.filter_algorithm)
        o.enable_filter = int(bool(self.enable_filter))
        o.enable_uniquify = int(bool(self.enable_uniquify))
        o.uniquify_algorithm = int(self.uniquify_algorithm)
        o.best_effort_uniquify = int(bool(self.best_effort_uniquify))
        o.uniquify_percent = float(self.uniquify_percent)
        o.engine_flags = int(self.engine_flags)
        o.advance_direction = int(self.advance_direction)
        o.max_iterations = int(self.max_iterations)
        object.__setattr__(self, "_cached", o)
        return o


class graph_t:
    """graph::graph_t with a CSR view (include/gunrock/graph/graph.hxx:53-339).
    Non-owning on the engine side; this object keeps the tensors alive."""

    def __init__(self, properties, tensors, context):
        self.properties = properties
        self._tensors = tensors
        self._ctx = context
        ro, ci, x = tensors
        self._V = int(ro.numel()) - 1
        self._E = int(ci.numel())
        self._h = C.c_void_p()
        _capi.check(_capi.lib().grx_graph_create_csr(
            context._h, self._V, self._E, C.c_void_p(ro.data_ptr()),
            C.c_void_p(ci.data_ptr()), C.c_void_p(x.data_ptr()) if x is not None else None,
            int(properties.directed), int(properties.weighted), int(properties.symmetric),
            C.byref(self._h)))

    def get_number_of_vertices(self):
        return self._V

    def get_number_of_edges(self):
        return self._E

    def is_directed(self):
        return self.properties.directed

    def is_symmetric(self):
        return self.properties.symmetric

    def is_weighted(self):
        return self.properties.weighted

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _capi.lib().grx_graph_destroy(self._h)
                self._h = None
        except Exception:
            pass


_default_ctx = {}


def _context(context, device=0):
    if context is not None:
        return context
    if device not in _default_ctx:
        _default_ctx[device] = multi_context_t(device)
    return _default_ctx[device]


def build_graph(properties, csr, context=None, device="cuda:0"):
    """graph::build<memory_space_t::device>(properties, csr)
    (include/gunrock/graph/build.hxx:29-36)."""
    import torch
    dev = torch.device(device)
    ctx = _context(context, dev.index or 0)
    return graph_t(properties, csr.to_device(device), ctx)


def _raw_current_stream(index):
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(index)  # no Stream object: ~0.3 us
    except AttributeError:
        return torch.cuda.current_stream(index).cuda_stream


def _order_after_producer(ctx, *tensors):
    """The engine works on ITS context's stream (non-blocking, like the reference's
    standard_context_t).  Work queued on torch's current stream that produces the
    tensors handed in (e.g. `torch.full` just before the call) must be finished first;
    when the context was created on that very stream, or that stream is idle, nothing
    needs to be done -- else the context's stream waits for an event recorded there
    (grx_context_order_after; the host does not block)."""
    for t in tensors:
        dev = getattr(t, "device", None)
        if dev is None or getattr(dev, "type", "") != "cuda":
            continue
        raw = _raw_current_stream(dev.index if dev.index is not None else 0)
        if ctx._stream_handle != raw:
            _capi.check(_capi.lib().grx_context_order_after(ctx._h, C.c_void_p(raw)))
        return


def _ptr(t, dtype_name):
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    iface = getattr(t, "__cuda_array_interface__", None) or getattr(t, "__hip_array_interface__", None)
    if iface:
        return C.c_void_p(iface["data"][0])
    raise RuntimeError("Object must be a PyTorch tensor or support "
                       "__cuda_array_interface__/__hip_array_interface__")


def bfs(graph, single_source, distances, predecessors=None, context=None, options=None):
    """gunrock.bfs(G, src, distances, predecessors, context, options) -> ms
    (python/src/gunrock/bindings.cu:233-266)."""
    ctx = context or graph._ctx
    ms = C.c_float(0)
    o = (options or options_t())._c()
    _order_after_producer(ctx, distances, predecessors)
    _capi.check(_capi.lib().grx_bfs(ctx._h, graph._h, int(single_source), C.byref(o),
                                    _ptr(distances, "int32"), _ptr(predecessors, "int32"),
                                    C.byref(ms)))
    return ms.value


def sssp(graph, single_source, distances, predecessors=None, context=None, options=None):
    """gunrock.sssp(...) -> ms (python/src/gunrock/bindings.cu:186-222)."""
    ctx = context or graph._ctx
    ms = C.c_float(0)
    o = (options or options_t())._c()
    _order_after_producer(ctx, distances, predecessors)
    _capi.check(_capi.lib().grx_sssp(ctx._h, graph._h, int(single_source), C.byref(o),
                                     _ptr(distances, "float32"), _ptr(predecessors, "int32"),
                                     C.byref(ms)))
    return ms.value


class bfs_param_t:
    def __init__(self, single_source, options=None):
        self.single_source = single_source
        self.options = options or options_t()


class sssp_param_t(bfs_param_t):
    pass


class pr_param_t:
    """pr::param_t (include/gunrock/algorithms/pr.hxx:19-27)."""

    def __init__(self, alpha=0.85, tol=1e-6, options=None):
        self.alpha = alpha
        self.tol = tol
        self.options = options or options_t()


class pr_result_t:
    """pr::result_t (pr.hxx:29-33): holds the output rank tensor."""

    def __init__(self, p):
        self.p = p
        self.iterations = 0


def pr_run(graph, param, result, context=None):
    """gunrock.pr_run(G, param, result, context) -> ms (bindings.cu:305-314)."""
    ctx = context or graph._ctx
    ms = C.c_float(0)
    it = C.c_int32(0)
    o = param.options._c()
    _order_after_producer(ctx, result.p)
    _capi.check(_capi.lib().grx_pr(ctx._h, graph._h, float(param.alpha), float(param.tol),
                                   C.byref(o), _ptr(result.p, "float32"), C.byref(it),
                                   C.byref(ms)))
    result.iterations = it.value
    return ms.value


def run_stats(context):
    s = grx_run_stats_t()
    _capi.check(_capi.lib().grx_get_run_stats(context._h, C.byref(s)))
    return {"edges_visited": s.edges_visited, "vertices_visited": s.vertices_visited,
            "search_depth": s.search_depth, "elapsed_ms": s.elapsed_ms, "aux": s.reserved}


def has_block_async():
    """True when the loaded library carries the block-asynchronous relaxation (gunrock_amd/libgrx_block.so; not the default)."""
    return bool(_capi.lib().grx_has_block_async())


def block_stats(context):
    """Statistics of the last block-asynchronous search on the context (road-like graphs, grx_block.hip);
    supersteps == 0: the last search took another path."""
    s = grx_block_stats_t()
    _capi.check(_capi.lib().grx_get_block_stats(context._h, C.byref(s)))
    return {k: getattr(s, k) for k, _ in grx_block_stats_t._fields_}


def level_profile(context, capacity=65536):
    arr = (grx_level_profile_t * capacity)()
    n = C.c_int32(0)
    _capi.check(_capi.lib().grx_get_level_profile(context._h, arr, capacity, C.byref(n)))
    return [{"frontier_size": arr[i].frontier_size, "edges": arr[i].edges,
             "advance_ms": arr[i].advance_ms, "other_ms": arr[i].other_ms,
             "bottom_up": arr[i].bottom_up, "bu_open": arr[i].bu_open, "bu_probes": arr[i].bu_probes}
            for i in range(min(n.value, capacity))]


def generate_rows(kind, n_vertices, n_entries, row_lo, row_hi, a=0.57, b=0.19, c=0.19, seed=42, in_rows=False):
    """The rows [row_lo, row_hi) of generate(kind, ...): same edges, same order, global
    column ids, n_vertices rows with the others empty -- the slice one rank owns.
    in_rows=True: the rows of the TRANSPOSE instead (in-edges of the owned vertices, global
    source ids) -- what the bottom-up step of a partitioned directed graph needs."""
    kinds = {"rmat": 0, "rmat_sym": 1}
    L = _capi.lib()
    h = C.c_void_p()
    fn = L.grx_host_csr_generate_in_rows if in_rows else L.grx_host_csr_generate_rows
    _capi.check(fn(kinds[kind], int(n_vertices), int(n_entries), float(a), float(b),
                   float(c), int(seed), int(row_lo), int(row_hi), C.byref(h)))
    try:
        ro, ci, x, props = _host_csr_to_numpy(h)
    finally:
        L.grx_host_csr_destroy(h)
    return props, csr_t()._set(ro, ci, x)


def generate(kind, n_vertices, n_entries=0, a=0.57, b=0.19, c=0.19, seed=42):
    """Seeded synthetic stand-ins for the BASELINE graphs.
    kind: 'rmat' (directed pattern), 'rmat_sym' (symmetric pattern), 'road'
    (lattice, p_keep=a, weighted iff c>0), 'rmat_deep' (directed R-MAT core + a long-tailed periphery: ~15 levels
    from a hub where the plain R-MAT has 7).  Returns (graph_properties_t, csr_t)."""
    kinds = {"rmat": 0, "rmat_sym": 1, "road": 2, "rmat_deep": 3}
    L = _capi.lib()
    h = C.c_void_p()
    _capi.check(L.grx_host_csr_generate(kinds[kind], int(n_vertices), int(n_entries),
                                        float(a), float(b), float(c), int(seed), C.byref(h)))
    try:
        ro, ci, x, props = _host_csr_to_numpy(h)
    finally:
        L.grx_host_csr_destroy(h)
    return props, csr_t()._set(ro, ci, x)
