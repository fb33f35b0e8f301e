#!/bin/bash
# Round-end evidence in one GPU call: rocprofv3 passes (kernel stats + PMC classes -> profiles/r2_bench_pmc.json),
# the GPU test suite, smoke(), the bench line (with traffic from the passes just taken), A/B runs.
# Outputs: gpurun_out/final_* and gpurun_out/r2_*; copy what is to be judged into profiles/.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles
bash tools/profile_r2.sh > gpurun_out/final_profile.log 2>&1
cp gpurun_out/r2_bench_pmc.json profiles/r2_bench_pmc.json
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/final_pytest_gpu.log)
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/final_smoke.log)
timeout 900 python bench.py > gpurun_out/final_bench.log 2>&1; echo "rc $?" >> gpurun_out/final_bench.log
if [ "${FINAL_SHORT:-0}" != 1 ]; then
  timeout 300 python tools/ab_bfs.py lj claim do > gpurun_out/final_ab_lj.log 2>&1
  timeout 300 python tools/ab_bfs.py kron claim do > gpurun_out/final_ab_kron.log 2>&1
  GRX_MID_DEBUG=1 timeout 500 python tools/ab_mid.py 3 > gpurun_out/final_ab_mid.log 2>&1
  for algo in bfs ssspu sssp; do
    extra=""; [ $algo = bfs ] && extra="0 merge_path forward"; [ $algo != bfs ] && extra="0 merge_path"
    GRX_LB_STRICT=1 timeout 300 python tools/run_algo.py $algo road 3 $extra > gpurun_out/final_road_${algo}_strict1.log 2>&1
  done
  GRX_BIN_DEBUG=1 timeout 200 python tools/bin_debug.py lj > gpurun_out/final_bin_debug_l1.log 2>&1
  GRX_BIN_DEBUG=2 timeout 200 python tools/bin_debug.py lj > gpurun_out/final_bin_debug_l2.log 2>&1
  if [ "${FINAL_GENERIC:-0}" = 1 ]; then
    timeout 400 python tools/pr_mfma_experiment.py > gpurun_out/final_pr_mfma.log 2>&1
    bash tools/bench_generic.sh > gpurun_out/final_generic.log 2>&1
  fi
  timeout 900 python tests/tools/bench_all.py bfs_lj bfs_kron bfs_road sssp_road ssspu_road pr_kron > gpurun_out/final_bench_all.log 2>&1
fi
tail -2 gpurun_out/final_pytest_gpu.log; tail -1 gpurun_out/final_smoke.log; tail -2 gpurun_out/final_bench.log | cut -c1-300
