"""Per-wave phase clocks of the second bottom-up body at one level (tuning aid):
    GRX_BU_DEBUG=<level> python tools/bu_debug.py [lj|kron|twitter] """
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
o = gr.options_t(advance_direction=gr.optimized)
for _ in range(3):
    gr.bfs(G, src, d, None, ctx, o)
ctx.synchronize()
L = _capi.lib()
L.grx_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
buf = np.zeros(8 * 16384, dtype=np.int64)
_capi.check(L.grx_debug_read(ctx._h, buf.ctypes.data, buf.size))
r = buf.reshape(-1, 8)
r = r[r[:, 7] == -3]
tick = 0.01  # us
m32 = (1 << 32) - 1
t0 = r[:, 1].min()
dur = (r[:, 2] - r[:, 1]) * tick
print("level %s of %s: %d waves, kernel span %.1f us, wave start %.1f..%.1f, wave busy mean %.1f / max %.1f us"
      % (os.environ.get("GRX_BU_DEBUG"), name, len(r), (r[:, 2].max() - t0) * tick, 0.0, (r[:, 1].max() - t0) * tick, dur.mean(), dur.max()))
cols = {"prologue (slot words)": r[:, 3] & m32, "compaction of open vertices": r[:, 0] >> 32, "probe (records wait + frontier words)": r[:, 4] & m32, "outputs": r[:, 4] >> 32,
        "deferred pass": r[:, 5] & m32, "tile emission": r[:, 5] >> 32, "late merge + leftovers + totals": r[:, 3] >> 32}
for k, v in cols.items():
    print("  %-38s mean %6.2f us  max %6.2f us" % (k, v.mean() * tick, v.max() * tick))
rounds = r[:, 6] & 0xffff
deferred = (r[:, 6] >> 16) & 0xffffff
drains = r[:, 6] >> 40
print("  groups per wave mean %.1f max %d; deferred entries per wave mean %.1f max %d; deferred passes mean %.2f"
      % (rounds.mean(), rounds.max(), deferred.mean(), deferred.max(), drains.mean()))
print("  per group: probe %.2f us, outputs %.2f us" % ((cols["probe (records wait + frontier words)"].sum() / max(1, rounds.sum())) * tick,
                                                       (cols["outputs"].sum() / max(1, rounds.sum())) * tick))
