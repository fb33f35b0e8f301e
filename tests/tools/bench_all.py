"""All BASELINE.json single-GPU configs in one run: ours (engine) and, when
oracle/_ref/libgunrock_ref_gpu.so is present, the reference's own GPU path on the same
arrays (test infrastructure; reporting only).  Prints one JSON object per config.
    python tests/tools/bench_all.py [bfs_lj] [sssp_road] [ssspu_road] [pr_kron] [bfs_road] [sssp_lj] [ssspu_lj] [sssp_kron]
        [ssspu_kron] [pr_lj] [bfs_kron] [bfs_twitter]      (ssspu = unit weights, what the reference loader makes of a pattern file)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gunrock_amd as gr  # noqa: E402
import oracle_lib as O  # noqa: E402
from bench import WORKLOADS, pair_hash_weights  # noqa: E402

which = sys.argv[1:] or ["bfs_lj", "sssp_road", "pr_kron"]
ctx = gr.multi_context_t(0)


def graph(name, weighted=False):
    wl = WORKLOADS[name]
    c_par = 1.0 if (weighted and wl["kind"] == "road") else wl["c"]
    props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], c_par, seed=42)
    if weighted and wl["kind"] != "road":
        csr.nonzero_values = pair_hash_weights(csr)  # the weights bench.py and the full-size tests draw
        csr._device = None
        props.weighted = True
    return wl, props, csr


def med(xs):
    return sorted(xs)[len(xs) // 2]


for item in which:
    algo, name = item.split("_")
    wl, props, csr = graph(name, weighted=(algo == "sssp"))
    if algo == "ssspu":  # unit weights: what the reference loader makes of a pattern .mtx (road_usa)
        algo = "sssp"
    g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
    V = g.n_vertices
    src = int(np.argmax(np.diff(g.row_offsets)))
    if name == "road":
        src = (4894 // 2) * 4894 + 4894 // 2
    G = gr.build_graph(props, csr, ctx)
    out = {"config": item, "workload": wl["name"], "V": V, "E": g.n_edges, "source": src}
    ref = O.ref_gpu() if O.have_ref_gpu() else None
    rh = ref.ref_gpu_graph_create(V, g.n_edges, g.row_offsets, g.column_indices, g.values) if ref else None
    if algo == "bfs":
        d = torch.empty(V, dtype=torch.int32, device="cuda")
        for label, o in (("ours_topdown", gr.options_t(advance_load_balance=gr.merge_path)),
                         ("ours_direction_optimized", gr.options_t(advance_load_balance=gr.merge_path,
                                                                    advance_direction=gr.optimized))):
            for _ in range(2):
                gr.bfs(G, src, d, None, ctx, o)
            ts = [gr.bfs(G, src, d, None, ctx, o) for _ in range(7)]
            st = gr.run_stats(ctx)
            out[label] = {"enact_ms": round(med(ts), 4), "mteps": round(st["edges_visited"] / (med(ts) * 1e3), 1),
                          "levels": st["search_depth"], "edges": st["edges_visited"]}
        mine = d.cpu().numpy()
        out["property_check_violations"] = int(O.check_bfs(g, src, mine))
        if ref:
            h = np.empty(V, np.int32)
            # the reference's own load-balance options without a filter (what bin/bfs_generic --advance_load_balance X runs
            # on OUR operators: tools/bench_generic.sh) and its tuned README command (merge_path + filter)
            for label, lb, flt in (("ref_gpu_default_block_mapped", 2, 0), ("ref_gpu_merge_path_filter", 4, 1),
                                   ("ref_gpu_merge_path_no_filter", 4, 0), ("ref_gpu_thread_mapped_no_filter", 0, 0)):
                if lb == 0 and name != "lj":
                    continue
                ts = [ref.ref_gpu_bfs(rh, src, lb, flt, 1, h) for _ in range(3 if lb else 1)]
                out[label] = {"enact_ms": round(min(ts), 3), "mteps": round(st["edges_visited"] / (min(ts) * 1e3), 1),
                              "equal_to_ours": bool(np.array_equal(h, mine))}
    elif algo == "sssp":
        d = torch.empty(V, dtype=torch.float32, device="cuda")
        o = gr.options_t(advance_load_balance=gr.merge_path)
        for _ in range(2):
            gr.sssp(G, src, d, None, ctx, o)
        ts = [gr.sssp(G, src, d, None, ctx, o) for _ in range(5)]
        st = gr.run_stats(ctx)
        out["ours"] = {"enact_ms": round(med(ts), 4), "mteps": round(st["edges_visited"] / (med(ts) * 1e3), 1),
                       "levels": st["search_depth"], "edges_relaxed": st["edges_visited"]}
        mine = d.cpu().numpy()
        out["property_check_violations"] = int(O.check_sssp(g, src, mine))
        if ref:
            h = np.empty(V, np.float32)
            ts = [ref.ref_gpu_sssp(rh, src, 2, h) for _ in range(2)]
            out["ref_gpu_default"] = {"enact_ms": round(min(ts), 3), "equal_to_ours": bool(np.array_equal(h, mine))}
    else:
        p = torch.empty(V, dtype=torch.float32, device="cuda")
        res = gr.pr_result_t(p)
        par = gr.pr_param_t(0.85, 1e-6)
        t0 = time.time()
        gr.pr_run(G, par, res, ctx)
        first = time.time() - t0
        ts = [gr.pr_run(G, par, res, ctx) for _ in range(5)]
        out["ours"] = {"enact_ms": round(med(ts), 4), "iterations": res.iterations,
                       "ms_per_iteration": round(med(ts) / max(1, res.iterations), 4),
                       "mteps": round(g.n_edges * res.iterations / (med(ts) * 1e3), 1),
                       "first_call_s_incl_transpose": round(first, 3),
                       "alg_GBps": round((8 * g.n_edges + 16 * V) * res.iterations / (med(ts) * 1e-3) / 1e9, 1)}
        mine = p.cpu().numpy()
        if ref:
            h = np.empty(V, np.float32)
            ts = [ref.ref_gpu_pr(rh, 0.85, 1e-6, h) for _ in range(2)]
            out["ref_gpu"] = {"enact_ms": round(min(ts), 3), "max_abs_diff_to_ours": float(np.abs(h - mine).max())}
    if ref:
        ref.ref_gpu_graph_destroy(rh)
    print(json.dumps(out), flush=True)
    del G
