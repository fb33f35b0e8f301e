"""A/B of the block-asynchronous path on the road stand-in (grx_block.hip): block size x bucket width, BFS and weighted SSSP,
against the level-synchronous kernels on the same arrays.   python tools/ab_block.py [side=4894] [reps=3]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gunrock_amd as gr  # noqa: E402
import oracle_lib as O  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4894
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = gr.multi_context_t(0)
src = (side // 2) * side + side // 2
KNOBS = ("GRX_BLOCK_NV", "GRX_BLOCK_NV_W", "GRX_BLOCK_DELTA", "GRX_BLOCK_DELTA_W", "GRX_BLOCK_WG_PER_CU")


def run(kind, props, csr, env, flags=0):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(dict(env, GRX_BLOCK="1"))
    G = gr.build_graph(props, csr, ctx)  # the device arrays are cached in csr; the handle (and its block structure) is new
    if kind == "bfs":
        d = torch.empty(csr.number_of_rows, dtype=torch.int32, device="cuda")
        f = lambda: gr.bfs(G, src, d, None, ctx, gr.options_t(advance_load_balance=gr.merge_path, engine_flags=flags))
    else:
        d = torch.empty(csr.number_of_rows, dtype=torch.float32, device="cuda")
        f = lambda: gr.sssp(G, src, d, None, ctx, gr.options_t(advance_load_balance=gr.merge_path, engine_flags=flags))
    t0 = time.perf_counter()
    f()
    ctx.synchronize()
    first = (time.perf_counter() - t0) * 1e3
    ts = [f() for _ in range(reps)]
    bs, st = gr.block_stats(ctx), gr.run_stats(ctx)
    return d.cpu().numpy(), min(ts), first, bs, st


for kind, weighted in (("bfs", False), ("sssp", True)):
    props, csr = gr.generate("road", side * side, 0, 0.602, 0.0, 1.0 if weighted else 0.0, seed=42)
    g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
    base, ms0, first0, _, st0 = run(kind, props, csr, {}, gr.FLAG_NO_BLOCK_ASYNC)
    viol = O.check_sssp(g, src, base) if weighted else O.check_bfs(g, src, base)
    print("%s side %d level-synchronous: %.3f ms (first call %.1f ms), oracle violations %d" % (kind, side, ms0, first0, viol), flush=True)
    if weighted:
        grid = [({"GRX_BLOCK_NV_W": str(nv), "GRX_BLOCK_DELTA_W": str(dl)}) for nv in (2048, 4096) for dl in (8, 16, 32, 64)]
    else:
        grid = [({"GRX_BLOCK_NV": str(nv), "GRX_BLOCK_DELTA": str(dl)}) for nv in (2048, 8192) for dl in (32, 64, 256, 1024)]
    grid += [dict(grid[len(grid) // 2], GRX_BLOCK_WG_PER_CU="1")]
    for env in grid:
        got, ms, first, bs, st = run(kind, props, csr, env)
        print("  %-60s %8.3f ms  supersteps %5d buckets %4d activations %7d relaxed %.2fx  blocks %d cross %.2f%%  build %.0f ms "
              "first call %.0f ms  equal %s" % (env, ms, bs["supersteps"], bs["buckets"], bs["activations"],
                                               bs["edges_relaxed"] / max(1, g.n_edges), bs["blocks"],
                                               100.0 * bs["cross_edges"] / g.n_edges, bs["build_ms"], first,
                                               bool(np.array_equal(got, base))), flush=True)
