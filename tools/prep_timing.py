"""Per-graph preprocessing, step by step (GRX_PREP_TIMING=1 makes the library print every step): first calls on fresh
handles of the LJ and kron stand-ins.   python tools/prep_timing.py [lj] [kron]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GRX_PREP_TIMING"] = "1"
import gunrock_amd as gr  # noqa: E402
import bench  # noqa: E402

ctx = gr.multi_context_t(0)
for name in (sys.argv[1:] or ["lj", "kron"]):
    props, csr, src, _ = bench.load_workload(gr, name)
    V = csr.number_of_rows
    csr.to_device("cuda:0")
    d = torch.empty(V, dtype=torch.int32, device="cuda")
    p = torch.empty(V, dtype=torch.float32, device="cuda")
    for what in ("bfs forward", "bfs direction-optimising", "pagerank"):
        G = gr.build_graph(props, csr, ctx)
        torch.cuda.synchronize()
        for rep in ("first call", "second call"):
            t0 = time.perf_counter()
            if what == "pagerank":
                gr.pr_run(G, gr.pr_param_t(0.85, 1e-6), gr.pr_result_t(p), ctx)
            else:
                gr.bfs(G, src, d, None, ctx, gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                                                          advance_direction=gr.optimized if "optim" in what else gr.forward))
            ctx.synchronize()
            sys.stderr.flush()
            print("== %s %s, %s: %.3f ms" % (name, what, rep, (time.perf_counter() - t0) * 1e3), flush=True)
        del G
