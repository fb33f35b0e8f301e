"""Round-5 A/B across LIBRARY BUILDS (one process per build: GRX_LIB_PATH selects gunrock_amd/libgrx_<tag>.so):
    GRX_LIB_PATH=gunrock_amd/libgrx_r4.so python tools/ab_r5.py [lj|kron|twitter] [reps] [bfs,do,ssspw,multi]
The round's kernel changes are compile-time (an `s_waitcnt` the compiler places, a ring of registers), so the baseline is a build of
the previous sources, not an environment switch.  Every line carries a CRC of the result so that two processes can be compared, the
wall time per call (K calls back to back, ASYNC_RETURN for the BFS), the device-clock enact() time of the last one and -- forward
BFS -- the per-level profile (level kernels + head, us) with the roofline fraction of the two fattest levels."""
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from gunrock_amd import _capi  # noqa: E402
from bench import WORKLOADS, pair_hash_weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
what = set((sys.argv[3] if len(sys.argv) > 3 else "bfs,do,ssspw").split(","))
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
print("lib", os.path.basename(_capi.LIB_PATH), "| workload", name, "V", csr.number_of_rows, "E", csr.number_of_nonzeros, "src", src, flush=True)


def crc(t):
    return "%08x" % (zlib.crc32(t.cpu().numpy().tobytes()) & 0xffffffff)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ctx.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        ctx.synchronize()
        t = (time.perf_counter() - t0) * 1e3 / n
        best = t if best is None else min(best, t)
    return best


def set_env(env):
    keep = ("GRX_LIB_PATH", "GRX_KEEP") + tuple(os.environ.get("GRX_KEEP", "").split(","))  # GRX_KEEP=VAR,VAR: switches of the whole process
    for k in [k for k in os.environ if k.startswith("GRX_") and k not in keep]:
        os.environ.pop(k)
    for k, v in (env or {}).items():
        os.environ[k] = str(v)


def bfs_line(G, d, label, direction, env=None, source=None):
    s = src if source is None else source
    set_env(env)
    o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                     advance_direction=direction, engine_flags=gr.FLAG_ASYNC_RETURN)
    step = timed(lambda: gr.bfs(G, s, d, None, ctx, o), reps)
    st = gr.run_stats(ctx)
    c0 = crc(d)
    po = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                      advance_direction=direction, engine_flags=gr.FLAG_PROFILE)
    best = None
    for _ in range(3):
        gr.bfs(G, s, d, None, ctx, po)
        prof = gr.level_profile(ctx)
        t = sum(l["advance_ms"] for l in prof)
        if best is None or t < best[0]:
            best = (t, prof)
    fat = sorted(best[1], key=lambda l: -l["edges"])[:2]
    frac = sum(12 * l["frontier_size"] + 12 * l["edges"] for l in fat) / max(1e-9, sum(l["advance_ms"] for l in fat) * 1e-3) / 8e12
    lv = " ".join("%d/%d:%s%.0f+h%.0f" % (l["frontier_size"], l["edges"], {0: "T", 1: "B", 2: "N", 3: "M"}.get(l.get("bottom_up"), "?"),
                                          l["advance_ms"] * 1e3, l["other_ms"] * 1e3) for l in best[1])
    print("%-44s step %.4f ms | enact %.4f | GTEPS %.1f | groups %d | fat frac %.3f | crc %s %s | %s"
          % (label, step, st["elapsed_ms"], st["edges_visited"] / (step * 1e6), int(st["aux"]), frac, c0, crc(d), lv), flush=True)
    return step


if what & {"bfs", "do", "multi"}:
    G = gr.build_graph(props, csr, ctx)
    d = torch.empty(G.get_number_of_vertices(), dtype=torch.int32, device="cuda")
    if "bfs" in what:
        bfs_line(G, d, "fwd default", gr.forward)
        bfs_line(G, d, "fwd exact schedule off (GRX_BIN_EXACT=0)", gr.forward, {"GRX_BIN_EXACT": 0})
        bfs_line(G, d, "fwd hints off (every group, all kernels)", gr.forward, {"GRX_BIN_EXACT": 0, "GRX_BIN_HINT": 0, "GRX_GROUP_HINT": 0})
        bfs_line(G, d, "fwd default again", gr.forward)
    if "bfs" in what:
        # the bin table is cut once per graph handle: a fresh handle per cut
        for label, env in (("fwd uniform bins of 32768 (GRX_BIN_USHIFT=15)", {"GRX_BIN_USHIFT": 15}),
                           ("fwd balanced bins + granule table (GRX_BIN_UNIFORM=0)", {"GRX_BIN_UNIFORM": 0})):
            set_env(env)
            G2 = gr.build_graph(props, csr, ctx)
            bfs_line(G2, d, label, gr.forward, env)
            del G2
        bfs_line(G, d, "fwd default, third time", gr.forward)
    if "do" in what:
        bfs_line(G, d, "DO  default", gr.optimized)
    if "multi" in what:
        # other sources of the giant component, each searched ONCE per visit (the schedule is trained on whatever came before)
        rng = np.random.default_rng(7)
        deg = np.diff(csr.row_offsets)
        cand = rng.permutation(np.nonzero(deg > 0)[0])[:8]
        o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                         advance_direction=gr.forward, engine_flags=gr.FLAG_ASYNC_RETURN)
        set_env(None)
        tot_t = tot_e = 0.0
        for rnd in range(3):
            for s in cand:
                ctx.synchronize()
                t0 = time.perf_counter()
                gr.bfs(G, int(s), d, None, ctx, o)
                ctx.synchronize()
                t = time.perf_counter() - t0
                if rnd > 0:
                    tot_t += t
                    tot_e += gr.run_stats(ctx)["edges_visited"]
        print("fwd 8 other sources, one search each per round: %.1f GTEPS (edges / time over 2 rounds)" % (tot_e / tot_t / 1e9), flush=True)
    del G, d
if "ssspw" in what:
    w = pair_hash_weights(csr)
    csr_w = gr.csr_t.from_arrays(csr.row_offsets, csr.column_indices, w)
    dd = torch.empty(csr.number_of_rows, dtype=torch.float32, device="cuda")
    set_env(None)
    Gw = gr.build_graph(gr.graph_properties_t(directed=True, weighted=True, symmetric=False), csr_w, ctx)
    step = timed(lambda: gr.sssp(Gw, src, dd, None, ctx, gr.options_t()), max(3, reps // 4))
    st = gr.run_stats(ctx)
    print("sssp U{1..1000} default                      step %.4f ms | iterations %d | relaxed %d | crc %s"
          % (step, st["search_depth"], st["edges_visited"], crc(dd)), flush=True)
