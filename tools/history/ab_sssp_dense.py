"""Weighted SSSP on the dense stand-ins (LJ / kron, U{1..1000} per-pair weights): plain label-correcting levels (the
default there) against the near-far schedule at several bucket widths.   python tools/ab_sssp_dense.py lj|kron [scale ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
props, csr, src, _ = bench.load_workload(gr, name)
csr.nonzero_values = bench.pair_hash_weights(csr)
csr._device = None
props.weighted = True
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
V = G.get_number_of_vertices()
d = torch.empty(V, dtype=torch.float32, device="cuda")


def run(label, flags, env):
    for k in ("GRX_NF_DELTA_SCALE",):
        os.environ.pop(k, None)
    os.environ.update(env)
    o = gr.options_t(advance_load_balance=gr.merge_path, engine_flags=flags)
    gr.sssp(G, src, d, None, ctx, o)
    ts = sorted(gr.sssp(G, src, d, None, ctx, o) for _ in range(5))
    st = gr.run_stats(ctx)
    return label, ts[2], st, d.cpu().numpy().copy()


ref = run("plain levels (default)", 0, {})
print("%-32s %.3f ms  iterations %d  relaxed %d" % (ref[0], ref[1], ref[2]["search_depth"], ref[2]["edges_visited"]), flush=True)
for sc in [float(x) for x in (sys.argv[2:] or ["0.01", "0.02", "0.05", "0.1", "0.2", "0.5"])]:
    r = run("near-far, delta x%g" % sc, gr.FLAG_SSSP_NEAR_FAR, {"GRX_NF_DELTA_SCALE": str(sc)})
    print("%-32s %.3f ms  iterations %d  relaxed %d  buckets %s  same %s"
          % (r[0], r[1], r[2]["search_depth"], r[2]["edges_visited"], r[2]["aux"], bool(np.array_equal(r[3], ref[3]))), flush=True)
