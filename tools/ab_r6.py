"""Round-6 A/B of the forward BFS on one stand-in under a list of ENVIRONMENT settings inside one process (the knobs are read per
search), best of 3 timings of K searches each, CRC of the depths, per-level kernel times of a profiled search:
    python tools/ab_r6.py [lj|kron|twitter] [K] "NAME=VAL,NAME=VAL" ...            ("-" = defaults)"""
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
settings = sys.argv[3:] or ["-"]
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
V = csr.number_of_rows
d = torch.empty(V, dtype=torch.int32, device="cuda:0")
print("workload", name, "V", V, "E", csr.number_of_nonzeros, "src", src, flush=True)
for setting in settings:
    touched = []
    if setting != "-":
        for kv in setting.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
            touched.append(k)
    G = gr.build_graph(props, csr, ctx, device="cuda:0")  # (a fresh handle per setting: hints and tables are per handle)
    o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact, advance_direction=gr.forward,
                     engine_flags=gr.FLAG_ASYNC_RETURN)
    for _ in range(4):
        gr.bfs(G, src, d, None, ctx, o)
    ctx.synchronize()
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            gr.bfs(G, src, d, None, ctx, o)
        ctx.synchronize()
        t = (time.perf_counter() - t0) * 1e3 / K
        best = t if best is None else min(best, t)
    crc = zlib.crc32(d.cpu().numpy().tobytes()) & 0xffffffff
    po = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact, advance_direction=gr.forward,
                      engine_flags=gr.FLAG_PROFILE)
    prof = None
    for _ in range(3):
        gr.bfs(G, src, d, None, ctx, po)
        p = gr.level_profile(ctx)
        if prof is None or sum(l["advance_ms"] for l in p) < sum(l["advance_ms"] for l in prof):
            prof = p
    fat = sorted(prof, key=lambda l: -l["edges"])[:2]
    frac = sum(12 * (l["frontier_size"] + l["edges"]) for l in fat) / (sum(l["advance_ms"] for l in fat) * 1e-3) / 8e12
    print("  %-40s %.4f ms/search | fat levels %s us frac %.4f | levels us %s | crc %08x" % (
        setting, best, " ".join("%.1f" % (l["advance_ms"] * 1e3) for l in fat), frac,
        " ".join("%.0f+%.0f" % (l["advance_ms"] * 1e3, l["other_ms"] * 1e3) for l in prof), crc), flush=True)
    for k in touched:
        del os.environ[k]
    del G
