"""Round-4 (second session) A/B of the host-side schedule changes, ONE process per graph:
  GRX_GROUP_HINT         paced BFS: enqueue as many launch groups as the previous search needed, then wait (run_levels hold_after);
                         weighted SSSP on a dense graph / PageRank: first blind batch = the previous run's group count
  GRX_FWD_SEED_IN_RESET  forward BFS: labels + visited bitmap + seed in one launch
  GRX_SOURCE_MAP         forward BFS: the source kernel writes level 1's chunk map and counters
    python tools/ab_r4b.py [lj|kron|twitter] [reps] [bfs,sssp,pr]
Every line: wall time per search (K searches back to back, ASYNC_RETURN for the BFS), the device-clock enact() time of the
last one, launch groups of the last one, result equal to the first configuration's."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS, pair_hash_weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
what = set((sys.argv[3] if len(sys.argv) > 3 else "bfs,sssp,pr").split(","))
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
KNOBS = ("GRX_GROUP_HINT", "GRX_FWD_SEED_IN_RESET", "GRX_SOURCE_MAP", "GRX_BIN_MIN_EDGES", "GRX_SOURCE_WG_PER_CU", "GRX_PACE_DEPTH",
         "GRX_BIN_SWEEP", "GRX_SW2_WG_PER_CU", "GRX_SW2_PARTS_PER_BIN", "GRX_SW2_ITEMS", "GRX_RBIN_MIN_EDGES", "GRX_RBIN_PARTS", "GRX_SSSP_BFS_ASYNC")
# argv[3] may also name "knobs": sweeps of EXISTING tuning knobs on the final sources (sweep geometry of the binned BFS levels,
# threshold / parts of the binned relaxation) instead of the session's own switches


def set_env(env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in (env or {}).items():
        os.environ[k] = str(v)


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


print("workload", name, "V", csr.number_of_rows, "E", csr.number_of_nonzeros, "src", src, flush=True)
if "bfs" in what:
    G = gr.build_graph(props, csr, ctx)
    V = G.get_number_of_vertices()
    d = torch.empty(V, dtype=torch.int32, device="cuda")
    ref = {}
    for direction, dname in ((gr.forward, "fwd"), (gr.optimized, "DO ")):
        o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                         advance_direction=direction, engine_flags=gr.FLAG_ASYNC_RETURN)
        for label, env in (("default (all on)", {}),
                           ("no group hint", {"GRX_GROUP_HINT": 0}),
                           ("no fused reset+seed", {"GRX_FWD_SEED_IN_RESET": 0}),
                           ("no source chunk map", {"GRX_SOURCE_MAP": 0}),
                           ("all off (previous sources' schedule)", {"GRX_GROUP_HINT": 0, "GRX_FWD_SEED_IN_RESET": 0, "GRX_SOURCE_MAP": 0}),
                           ("default again", {}),
                           # existing knobs, measured here because they are free to try
                           ("binned from 2^19 out-edges (default 2^20)", {"GRX_BIN_MIN_EDGES": 1 << 19}),
                           ("source level: 1 workgroup per CU (default 4)", {"GRX_SOURCE_WG_PER_CU": 1}),
                           ("source level: 8 workgroups per CU", {"GRX_SOURCE_WG_PER_CU": 8}),
                           ("pace depth 1 (default 2)", {"GRX_PACE_DEPTH": 1}),
                           ("pace depth 3", {"GRX_PACE_DEPTH": 3}),
                           ("default, third time", {})):
            if dname == "DO " and "BIN_MIN" in "".join(env):
                continue
            if dname == "DO " and ("FWD_SEED" in "".join(env) or "SOURCE_MAP" in "".join(env)) and len(env) == 1:
                continue  # forward-only knobs
            set_env(env)
            step = timed(lambda: gr.bfs(G, src, d, None, ctx, o), reps)
            st = gr.run_stats(ctx)
            h = d.cpu().numpy()
            ref.setdefault("bfs", h.copy())
            print("%s %-40s step %.4f ms | enact %.4f | GTEPS %.1f | groups %d | same %s"
                  % (dname, label, step, st["elapsed_ms"], st["edges_visited"] / (step * 1e6), int(st["aux"]),
                     bool(np.array_equal(h, ref["bfs"]))), flush=True)
    del G, d
if "knobs" in what:
    G = gr.build_graph(props, csr, ctx)
    V = G.get_number_of_vertices()
    d = torch.empty(V, dtype=torch.int32, device="cuda")
    o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                     advance_direction=gr.forward, engine_flags=gr.FLAG_ASYNC_RETURN)
    refk = None
    for label, env in (("default (sweep 3: 1024 threads, 1 per CU, nb + 160 items)", {}),
                       ("sweep 3, 300 items", {"GRX_SW2_ITEMS": 300}),
                       ("sweep 3, 400 items", {"GRX_SW2_ITEMS": 400}),
                       ("sweep 3, 512 items", {"GRX_SW2_ITEMS": 512}),
                       ("sweep 2 (512 threads, 2 per CU, 2 parts per bin)", {"GRX_BIN_SWEEP": 2}),
                       ("sweep 2, 3 per CU, 5 parts per bin", {"GRX_BIN_SWEEP": 2, "GRX_SW2_WG_PER_CU": 3, "GRX_SW2_PARTS_PER_BIN": 5}),
                       ("sweep 2, 3 per CU, 8 parts per bin", {"GRX_BIN_SWEEP": 2, "GRX_SW2_WG_PER_CU": 3, "GRX_SW2_PARTS_PER_BIN": 8}),
                       ("sweep 2, 4 per CU, 12 parts per bin", {"GRX_BIN_SWEEP": 2, "GRX_SW2_WG_PER_CU": 4, "GRX_SW2_PARTS_PER_BIN": 12}),
                       ("default again", {})):
        set_env(env)
        step = timed(lambda: gr.bfs(G, src, d, None, ctx, o), reps)
        st = gr.run_stats(ctx)
        h = d.cpu().numpy()
        refk = h.copy() if refk is None else refk
        print("fwd %-56s step %.4f ms | enact %.4f | GTEPS %.1f | same %s"
              % (label, step, st["elapsed_ms"], st["edges_visited"] / (step * 1e6), bool(np.array_equal(h, refk))), flush=True)
    del G, d
    w = pair_hash_weights(csr)
    csr_w = gr.csr_t.from_arrays(csr.row_offsets, csr.column_indices, w)
    dd = torch.empty(csr.number_of_rows, dtype=torch.float32, device="cuda")
    refw = None
    for label, env in (("default (binned from 2^20 relaxed edges, parts = CUs)", {}),
                       ("binned from 2^19", {"GRX_RBIN_MIN_EDGES": 1 << 19}),
                       ("binned from 2^21", {"GRX_RBIN_MIN_EDGES": 1 << 21}),
                       ("binned from 2^22", {"GRX_RBIN_MIN_EDGES": 1 << 22}),
                       ("binned from 2^23", {"GRX_RBIN_MIN_EDGES": 1 << 23}),
                       ("parts 128", {"GRX_RBIN_PARTS": 128}),
                       ("parts 512", {"GRX_RBIN_PARTS": 512}),
                       ("default again", {})):
        set_env(env)
        # a fresh handle per line: the launch groups that carry the scatter / sweep kernels are OR-ed over a handle's searches
        Gw = gr.build_graph(gr.graph_properties_t(directed=True, weighted=True, symmetric=False), csr_w, ctx)
        step = timed(lambda: gr.sssp(Gw, src, dd, None, ctx, gr.options_t()), max(3, reps // 4))
        st = gr.run_stats(ctx)
        h = dd.cpu().numpy()
        refw = h.copy() if refw is None else refw
        print("sssp U{1..1000} %-44s step %.4f ms | iterations %d | relaxed %d | same %s"
              % (label, step, st["search_depth"], st["edges_visited"], bool(np.array_equal(h, refw))), flush=True)
        del Gw
    del dd
if "unit" in what:
    # SSSP with the weights the reference's loader gives a pattern file (all 1.0): the BFS engine + one conversion pass
    Gu = gr.build_graph(props, csr, ctx)
    du = torch.empty(csr.number_of_rows, dtype=torch.float32, device="cuda")
    refu = None
    for label, env in (("default (inner search returns on the published end)", {}), ("inner search blocks (before)", {"GRX_SSSP_BFS_ASYNC": 0}),
                       ("default again", {})):
        set_env(env)
        step = timed(lambda: gr.sssp(Gu, src, du, None, ctx, gr.options_t()), reps)
        st = gr.run_stats(ctx)
        h = du.cpu().numpy()
        refu = h.copy() if refu is None else refu
        print("sssp all 1.0 %-52s step %.4f ms | enact %.4f | depth %d | same %s"
              % (label, step, st["elapsed_ms"], st["search_depth"], bool(np.array_equal(h, refu))), flush=True)
    del Gu, du
if "sssp" in what:
    w = pair_hash_weights(csr)
    Gw = gr.build_graph(gr.graph_properties_t(directed=True, weighted=True, symmetric=False),
                        gr.csr_t.from_arrays(csr.row_offsets, csr.column_indices, w), ctx)
    dd = torch.empty(csr.number_of_rows, dtype=torch.float32, device="cuda")
    refw = None
    for label, env in (("default (first batch = previous depth)", {}), ("no group hint (4 + 8 + 16 ...)", {"GRX_GROUP_HINT": 0}),
                       ("default again", {})):
        set_env(env)
        step = timed(lambda: gr.sssp(Gw, src, dd, None, ctx, gr.options_t()), max(3, reps // 4))
        st = gr.run_stats(ctx)
        h = dd.cpu().numpy()
        refw = h.copy() if refw is None else refw
        print("sssp U{1..1000} %-40s step %.4f ms | enact %.4f | iterations %d | same %s"
              % (label, step, st["elapsed_ms"], st["search_depth"], bool(np.array_equal(h, refw))), flush=True)
    del Gw, dd
if "pr" in what:
    Gp = gr.build_graph(props, csr, ctx)
    p = torch.empty(csr.number_of_rows, dtype=torch.float32, device="cuda")
    res = gr.pr_result_t(p)
    par = gr.pr_param_t(0.85, 1e-6)
    refp = None
    for label, env in (("default (first batch = previous iterations + 1)", {}), ("no hint (4 + 8 ...)", {"GRX_GROUP_HINT": 0}),
                       ("default again", {})):
        set_env(env)
        step = timed(lambda: gr.pr_run(Gp, par, res, ctx), max(3, reps // 4))
        h = p.cpu().numpy()
        refp = h.copy() if refp is None else refp
        print("pr %-48s step %.4f ms | iterations %d | same %s"
              % (label, step, res.iterations, bool(np.array_equal(h, refp))), flush=True)
