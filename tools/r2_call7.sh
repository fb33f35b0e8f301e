#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_bfs_gpu.py tests/test_sssp_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q -k "not full_size" > gpurun_out/c7_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c7_pytest.log)
tail -5 gpurun_out/c7_pytest.log
for m in 1 0; do
  GRX_MID=$m timeout 200 python tools/run_algo.py bfs road 3 0 merge_path forward > gpurun_out/c7_bfs_road_mid$m.log 2>&1; tail -1 gpurun_out/c7_bfs_road_mid$m.log | cut -c1-300
done
GRX_MID=1 timeout 600 python bench.py --only sssp --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/c7_bench_sssp.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c7_bench_sssp.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    for k in ('unit_weights','weighted_1_1000'):
        s=d['sssp'][k]; print(k, s['ms_per_step'], s['mteps'], s['us_per_iteration'], s['iterations'])
else:
    print(open('gpurun_out/c7_bench_sssp.log').read()[-2000:])
PY
(timeout 600 python -m pytest tests/test_sssp_gpu.py tests/test_bfs_gpu.py -m gpu -x -q -k "full_size" > gpurun_out/c7_pytest_big.log 2>&1; echo "pytest rc $?" >> gpurun_out/c7_pytest_big.log); tail -3 gpurun_out/c7_pytest_big.log
