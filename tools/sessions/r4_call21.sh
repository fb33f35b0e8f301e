#!/bin/bash
# round 4, call 21: device-side COO -> CSR (grx_csr_from_coo_device): Python and C++ tests, timing against the host builder
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sort_gpu.py tests/test_cli.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r4c21_pytest.log; cat gpurun_out/r4c21_pytest.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c21_coo_timing.log
import time, numpy as np, torch
import gunrock_amd as gr
from bench import WORKLOADS
wl = WORKLOADS["lj"]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
V, E = csr.number_of_rows, csr.number_of_nonzeros
rows = np.repeat(np.arange(V, dtype=np.int32), np.diff(csr.row_offsets))
perm = np.random.default_rng(1).permutation(E)
I, J = rows[perm].copy(), csr.column_indices[perm].copy()
X = np.ones(E, dtype=np.float32)
coo = gr.coo_t(); coo.number_of_rows = coo.number_of_columns = V; coo.number_of_nonzeros = E
coo.row_indices, coo.column_indices, coo.nonzero_values = I, J, X
t0 = time.perf_counter(); want = gr.csr_t().from_coo(coo); t_host = time.perf_counter() - t0
ctx = gr.multi_context_t(0)
dI, dJ, dX = torch.from_numpy(I).cuda(), torch.from_numpy(J).cuda(), torch.from_numpy(X).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ro, ci, x = gr.csr_t.from_coo_device(dI, dJ, dX, V, ctx)
    torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
same = bool(np.array_equal(ro.cpu().numpy(), want.row_offsets) and np.array_equal(ci.cpu().numpy(), want.column_indices))
print("COO -> CSR, LJ stand-in shuffled (%d rows, %d entries): host builder %.1f ms, device %.2f ms, identical %s" % (V, E, t_host * 1e3, t_dev * 1e3, same))
PY
