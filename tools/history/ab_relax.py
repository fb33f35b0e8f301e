"""Weighted SSSP on the dense stand-ins (LJ / kron, U{1..1000} per-pair weights): the relax-per-edge levels (round 3,
GRX_FLAG_SSSP_NO_BINS) against the binned relaxation of the fat levels (grx_relax.hpp), with the per-level profile.
    python tools/ab_relax.py lj|kron [VAR=value,VAR=value ...]"""
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
    for k in [k for k in os.environ if k.startswith("GRX_RBIN_")]:
        os.environ.pop(k, None)
    os.environ.update(env)
    o = gr.options_t(advance_load_balance=gr.merge_path, engine_flags=flags)
    first = gr.sssp(G, src, d, None, ctx, o)
    gr.sssp(G, src, d, None, ctx, o)
    ts = sorted(gr.sssp(G, src, d, None, ctx, o) for _ in range(7))
    st = gr.run_stats(ctx)
    out = d.cpu().numpy().copy()
    o.engine_flags = flags | gr.FLAG_PROFILE
    gr.sssp(G, src, d, None, ctx, o)
    prof = gr.level_profile(ctx)
    lv = " ".join("%d/%d:%s%.0f+h%.0f" % (r["frontier_size"], r["edges"], {0: "T", 2: "R", 3: "M"}.get(r["bottom_up"], "?"),
                                          r["advance_ms"] * 1e3, r["other_ms"] * 1e3) for r in prof)
    print("%-58s %.3f ms (first %.2f)  levels %d  relaxed %d | %s" % (label, ts[3], first, st["search_depth"], st["edges_visited"], lv), flush=True)
    return out


ref = run("relax per edge (round 3)", gr.FLAG_SSSP_NO_BINS, {})
# variants: comma-separated environment settings, e.g.  GRX_RBIN_MIN_EDGES=1048576,GRX_RBIN_PARTS=256
for spec in (sys.argv[2:] or ["", "GRX_RBIN_MIN_EDGES=1048576", "GRX_RBIN_PARTS=256", "GRX_RBIN_MIN_EDGES=1048576,GRX_RBIN_PARTS=256"]):
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    r = run("binned %s" % (spec or "(defaults)"), 0, env)
    print("   same as relax-per-edge: %s" % bool(np.array_equal(r, ref)), flush=True)
