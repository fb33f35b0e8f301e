#!/bin/bash
# Round-end evidence in one GPU call: gpu tests, smoke(), rocprofv3 passes of bench.py, the bench
# line itself, every BASELINE config next to the reference's own GPU path.  Outputs: gpurun_out/final_*
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/final_pytest_gpu.log)
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/final_smoke.log)
bash tools/profile.sh bench_final python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/final_profile_sh.log 2>&1
python tools/pmc_json.py gpurun_out/prof_bench_final > gpurun_out/final_bench_pmc.json 2> gpurun_out/final_pmc_json.err
cp gpurun_out/prof_bench_final/summary.md gpurun_out/final_summary.md
cp gpurun_out/prof_bench_final/kt/p_kernel_stats.csv gpurun_out/final_kernel_stats.csv
rm -rf gpurun_out/prof_bench_final
cp gpurun_out/final_bench_pmc.json profiles/r1_bench_pmc.json
timeout 300 python bench.py > gpurun_out/final_bench.log 2>&1
[ "${FINAL_SHORT:-0}" = 1 ] || timeout 200 python bench.py --workload kron --no-cpu-baseline > gpurun_out/final_bench_kron.log 2>&1
[ "${FINAL_SHORT:-0}" = 1 ] || timeout 300 python bench.py --workload road --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/final_bench_road.log 2>&1
[ "${FINAL_SHORT:-0}" = 1 ] || timeout 600 python tests/tools/bench_all.py bfs_lj bfs_kron bfs_road sssp_lj sssp_road ssspu_road pr_kron > gpurun_out/final_bench_all.log 2>&1
tail -2 gpurun_out/final_pytest_gpu.log; tail -2 gpurun_out/final_smoke.log; tail -1 gpurun_out/final_bench.log | cut -c1-160
