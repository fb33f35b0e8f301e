#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_distributed.py tests/test_sssp_gpu.py tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q -k "not full_size" > gpurun_out/c10_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c10_pytest.log)
tail -4 gpurun_out/c10_pytest.log
timeout 600 python tools/pr_mfma_experiment.py > gpurun_out/c10_mfma.log 2>&1; tail -1 gpurun_out/c10_mfma.log | cut -c1-1500
for strict in 0 1; do
  GRX_LB_STRICT=$strict timeout 200 python tools/run_algo.py bfs road 3 0 merge_path forward > gpurun_out/c10_bfs_road_strict$strict.log 2>&1; tail -1 gpurun_out/c10_bfs_road_strict$strict.log | cut -c1-260
done
timeout 300 python tools/ab_bfs.py kron bin2 > gpurun_out/c10_ab_kron.log 2>&1; grep "TD " gpurun_out/c10_ab_kron.log | cut -c1-300
timeout 300 python tools/ab_bfs.py lj bin2 > gpurun_out/c10_ab_lj.log 2>&1; grep "TD " gpurun_out/c10_ab_lj.log | cut -c1-300
GRX_MID=1 timeout 600 python bench.py --only sssp --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/c10_bench_sssp.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c10_bench_sssp.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    for k in ('unit_weights','weighted_1_1000'):
        s=d['sssp'][k]; print(k, s['ms_per_step'], s['mteps'], s['us_per_iteration'], s['iterations'])
else:
    print(open('gpurun_out/c10_bench_sssp.log').read()[-1500:])
PY
