"""Round-5 A/B of the PageRank run across library builds (GRX_LIB_PATH):  python tools/ab_pr5.py [kron|lj] [GRX_PR_WG_PER_CU values ...]
One line per setting: wall ms per run (best of 5), iterations, ms per iteration, CRC of the ranks."""
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
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "kron"
settings = sys.argv[2:] or [""]
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
p = torch.empty(csr.number_of_rows, dtype=torch.float32, device="cuda")
par = gr.pr_param_t(0.85, 1e-6)
res = gr.pr_result_t(p)
for setting in settings:
    # a setting: "" (defaults) | "<workgroups per CU>" | "hot=<entries>[,wg=<n>]"
    os.environ.pop("GRX_PR_WG_PER_CU", None)
    os.environ.pop("GRX_PR_HOT", None)
    for part in [x for x in setting.split(",") if x]:
        if part.startswith("hot="):
            os.environ["GRX_PR_HOT"] = part[4:]
        elif part.startswith("wg="):
            os.environ["GRX_PR_WG_PER_CU"] = part[3:]
        else:
            os.environ["GRX_PR_WG_PER_CU"] = part
    gr.pr_run(G, par, res, ctx)
    ctx.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        gr.pr_run(G, par, res, ctx)
        ctx.synchronize()
        t = (time.perf_counter() - t0) * 1e3
        best = t if best is None else min(best, t)
    print("lib %-16s %-5s setting %-16s run %.4f ms | iterations %d | %.4f ms per iteration | crc %08x"
          % (os.path.basename(_capi.LIB_PATH), name, setting or "resident", best, res.iterations, best / max(1, res.iterations),
             zlib.crc32(p.cpu().numpy().tobytes()) & 0xffffffff), flush=True)
