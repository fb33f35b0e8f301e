"""Per-workgroup timeline of the binned kernels of one BFS level (tuning aid):
    GRX_BIN_DEBUG=<level> python tools/bin_debug.py [lj|kron] """
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from gunrock_amd import _capi  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "lj"
wl = WORKLOADS[name]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
src = int(np.argmax(np.diff(csr.row_offsets)))
ctx = gr.multi_context_t(0)
G = gr.build_graph(props, csr, ctx)
d = torch.empty(G.get_number_of_vertices(), dtype=torch.int32, device="cuda")
o = gr.options_t(advance_direction=gr.forward)
for _ in range(3):
    gr.bfs(G, src, d, None, ctx, o)
ctx.synchronize()
L = _capi.lib()
L.grx_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
buf = np.zeros(8 * 16384, dtype=np.int64)
_capi.check(L.grx_debug_read(ctx._h, buf.ctypes.data, buf.size))
rec = buf.reshape(-1, 8)
for phase, lo in (("scatter", 0), ("claim", 4096)):
    r = rec[lo:lo + 4096]
    r = r[r[:, 3] != 0]
    if len(r) == 0:
        print(phase, "no records")
        continue
    t0 = r[:, 2].min()
    tick_us = 0.01  # wall_clock64: 100 MHz
    print("%s: %d workgroups, span %.1f us" % (phase, len(r), (r[:, 3].max() - t0) * tick_us))
    if phase == "claim" and (r[:, 7] == -2).all():
        # the slowest workgroups of the second sweep, phase by phase (a sweep is as slow as its slowest item)
        m32 = (1 << 32) - 1
        for q in r[np.argsort(-(r[:, 3] - r[:, 2]))[:6]]:
            print("  slow wg: xcc %d items %d entries %7d start %.1f busy %.1f us | stream %.1f merge %.1f expand %.1f emit %.1f"
                  % (q[0], q[1], q[4], (q[2] - t0) * tick_us, (q[3] - q[2]) * tick_us, (q[5] & m32) * tick_us, (q[5] >> 32) * tick_us,
                     (q[6] & m32) * tick_us, (q[6] >> 32) * tick_us))
        one = r[r[:, 1] > 0]
        print("  items %d, entries per item: mean %.0f max %d; busy of workgroups with an item: mean %.1f max %.1f us"
              % (len(one), one[:, 4].mean(), one[:, 4].max(), ((one[:, 3] - one[:, 2]) * tick_us).mean(), ((one[:, 3] - one[:, 2]) * tick_us).max()))
    for x in sorted(set(r[:, 0].tolist())):
        q = r[r[:, 0] == x]
        dur = (q[:, 3] - q[:, 2]) * tick_us
        extra = ""
        if phase == "scatter" and (q[:, 7] == -2).all():  # second scatter: wave 0's clock per phase
            nb = max(1, q[:, 1].sum())
            m20 = (1 << 20) - 1
            parts = (q[:, 4] & m20, (q[:, 4] >> 20) & m20, q[:, 4] >> 40, q[:, 5] & m20, (q[:, 5] >> 20) & m20, q[:, 5] >> 40,
                     q[:, 6] & m20, (q[:, 6] >> 20) & m20)
            extra = (" | per batch us: front+scan %.2f prefix+marks %.2f running-max %.2f owners+ci-arrive %.2f table+hist %.2f "
                     "reserve+bin-scan %.2f sort %.2f copy-out %.2f" % tuple(x.sum() * tick_us / nb for x in parts))
        elif phase == "claim" and (q[:, 7] == -2).all():  # second sweep
            m32 = (1 << 32) - 1
            ni = max(1, q[:, 1].sum())
            extra = (" entries %d (max %d) | per item us: stream %.1f merge %.1f expand %.1f emit %.1f"
                     % (q[:, 4].sum(), q[:, 4].max(), (q[:, 5] & m32).sum() * tick_us / ni, (q[:, 5] >> 32).sum() * tick_us / ni,
                        (q[:, 6] & m32).sum() * tick_us / ni, (q[:, 6] >> 32).sum() * tick_us / ni))
        elif phase == "scatter":
            nb = max(1, q[:, 1].sum())
            m20 = (1 << 20) - 1
            parts = (q[:, 4], q[:, 5] >> 40, (q[:, 5] >> 20) & m20, q[:, 6] >> 40, q[:, 5] & m20, q[:, 6] & ((1 << 40) - 1), q[:, 7])
            extra = " | per batch us: front %.2f scans %.2f search %.2f ci-arrive %.2f hist %.2f reserve+scan+sort %.2f copy-out %.2f" % tuple(
                x.sum() * tick_us / nb for x in parts)
        if phase == "claim" and (q[:, 7] == -2).all():
            pass
        elif phase == "claim" and os.environ.get("GRX_BIN_CLAIM", "3") == "2":
            extra = " entries %d bitmap words %d queue items %d dense %s" % (q[:, 4].sum(), q[:, 5].sum(), q[0, 6], set(q[:, 7].tolist()))
        elif phase == "claim":
            # sweep claim: [6] = clock after the candidate pass, [7] = after the word claims (single-item workgroups)
            one = q[q[:, 1] == 1]
            if len(one):
                extra = " entries %d (max %d) words %d | 1-item wgs: stream %.1f claim %.1f expand %.1f us (means), slowest %.1f us with %d entries / %d words" % (
                    q[:, 4].sum(), q[:, 4].max(), q[:, 5].sum(), ((one[:, 6] - one[:, 2]) * tick_us).mean(),
                    ((one[:, 7] - one[:, 6]) * tick_us).mean(), ((one[:, 3] - one[:, 7]) * tick_us).mean(),
                    ((one[:, 3] - one[:, 2]) * tick_us).max(), one[np.argmax(one[:, 3] - one[:, 2]), 4], one[np.argmax(one[:, 3] - one[:, 2]), 5])
        print("  xcc %d: %4d wgs, work units %6d, start %.1f..%.1f end %.1f..%.1f us, busy mean %.1f max %.1f%s"
              % (x, len(q), q[:, 1].sum(), (q[:, 2].min() - t0) * tick_us, (q[:, 2].max() - t0) * tick_us,
                 (q[:, 3].min() - t0) * tick_us, (q[:, 3].max() - t0) * tick_us, dur.mean(), dur.max(), extra))
