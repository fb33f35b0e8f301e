import sys, time, json, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import gunrock_amd as gr
from bench import WORKLOADS
wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "lj"]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0); G = gr.build_graph(props, csr, ctx)
d = torch.empty(G.get_number_of_vertices(), dtype=torch.int32, device="cuda")
ref = None
for variant in (0, 1, 4):
    o = gr.options_t(advance_load_balance=gr.merge_path, engine_flags=((variant & 3) << 8),
                     advance_direction=gr.optimized if variant == 4 else gr.forward)
    for _ in range(3): gr.bfs(G, src, d, None, ctx, o)
    ts = []
    for _ in range(10):
        ts.append(gr.bfs(G, src, d, None, ctx, o))
    st = gr.run_stats(ctx)
    h = d.cpu().numpy()
    if ref is None: ref = h
    po = gr.options_t(advance_load_balance=gr.merge_path, engine_flags=((variant & 3) << 8) | gr.FLAG_PROFILE,
                      advance_direction=gr.optimized if variant == 4 else gr.forward)
    gr.bfs(G, src, d, None, ctx, po)
    prof = gr.level_profile(ctx)
    print("variant", variant, "enact ms min %.3f med %.3f" % (min(ts), sorted(ts)[5]), "same", bool((h == ref).all()),
          "aux", st["aux"], "levels", [(l["frontier_size"], l["edges"], round(l["advance_ms"], 3)) for l in prof])
