"""Forward searches from OTHER sources than the bench's hub, one each, after the handle's hints were trained on the hub (the
protocol of bench.py's multi_source section):  python tools/ms_trace.py [lj] [src ...]      (default: the sources named below)
Prints one line per search (wall ms between two synchronisations, depth, GTEPS); under `rocprofv3 --kernel-trace` the kernel
sequence of every search is printed by tools/ms_trace.sh."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
# multi_source sources of the LJ stand-in with: a ~10-vertex level of 10-32 k edges (3761365, 1382913, 1455877), an early
# 1.7 M-edge level (269895), early 0.7-0.9 M-edge levels (4424746, 4042365), none of these (4580952, 4350118)
srcs = [int(x) for x in sys.argv[2:]] or [3761365, 1382913, 1455877, 269895, 4424746, 4042365, 4580952, 4350118]
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
hub = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
d = torch.empty(G.get_number_of_vertices(), dtype=torch.int32, device="cuda")
o = gr.options_t(advance_load_balance=gr.merge_path, enable_filter=True, filter_algorithm=gr.compact,
                 advance_direction=gr.optimized if os.environ.get("MS_DIR") == "do" else gr.forward, engine_flags=gr.FLAG_ASYNC_RETURN)
for _ in range(3):
    gr.bfs(G, hub, d, None, ctx, o)
ctx.synchronize()
# A/B inside one process (the engine reads its knobs per call): the rules of round 5's last session -- no heavy tile in the
# many-levels body, early levels binned from 2^20 edges -- against the sources without them
OLD = {"GRX_MID_TILE_E": "0", "GRX_BIN_EARLY_DIV": "1"}
ROUNDS = [("new", {}), ("old", OLD), ("new", {}), ("old", OLD)]
if os.environ.get("MS_DIR") == "do":  # direction-optimising searches: the two rules do not apply, one setting, two rounds
    ROUNDS = [("do", {}), ("do", {})]
for rep, (tag, env) in enumerate(ROUNDS):
    for k in OLD:
        os.environ.pop(k, None)
    os.environ.update(env)
    tot_e, tot_t = 0, 0.0
    for s in srcs:
        t1 = time.perf_counter()
        gr.bfs(G, s, d, None, ctx, o)
        ctx.synchronize()
        dt = time.perf_counter() - t1
        st = gr.run_stats(ctx)
        crc = int(torch.sum(d.to(torch.int64) * (d != 2147483647)).item()) & 0xffffffff
        print("round %d %s src %8d  %.4f ms  depth %2d  %6.1f GTEPS  sum %08x" % (rep, tag, s, dt * 1e3, st["search_depth"], st["edges_visited"] / (dt * 1e9), crc))
        tot_e += st["edges_visited"]
        tot_t += dt
    print("round %d %s: %.1f GTEPS over %d sources" % (rep, tag, tot_e / (tot_t * 1e9), len(srcs)))
