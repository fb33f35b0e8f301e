#!/bin/bash
# Round-3 evidence in one GPU call, on the FINAL sources: rocprofv3 passes (kernel stats + PMC classes -> r3_bench_pmc.json, whose
# source_sha bench.py checks before attaching `traffic`), the GPU test suite, smoke(), the bench line, every BASELINE config beside
# the reference's GPU path, the binned-level timelines, the A/B of the scatter / sweep versions, the generic operators, a fuzz sweep.
# Outputs under gpurun_out/; tools/collect_evidence_r3.sh copies what is to be judged into profiles/.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles
export TMPDIR=/tmp
bash tools/profile_r3.sh > gpurun_out/final_profile.log 2>&1
cp gpurun_out/r3_bench_pmc.json profiles/r3_bench_pmc.json
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/final_pytest_gpu.log)
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/final_smoke.log)
timeout 900 python bench.py > gpurun_out/final_bench.log 2>&1; echo "rc $?" >> gpurun_out/final_bench.log
if [ "${FINAL_SHORT:-0}" != 1 ]; then
  timeout 300 python tools/ab_r3.py lj 20 > gpurun_out/final_ab_lj.log 2>&1
  timeout 300 python tools/ab_r3.py kron 10 > gpurun_out/final_ab_kron.log 2>&1
  timeout 400 python tools/ab_r3.py twitter 5 > gpurun_out/final_ab_twitter.log 2>&1
  for g in lj kron twitter; do timeout 400 python tools/ab_bu.py $g 20 2>&1 | grep -v amdgpu.ids > gpurun_out/final_ab_bu_$g.log; done
  for l in 1 2 3; do GRX_BU_DEBUG=$l timeout 200 python tools/bu_debug.py lj 2>&1 | grep -v amdgpu.ids; done > gpurun_out/final_bu_debug_lj.log
  GRX_BIN_DEBUG=1 timeout 200 python tools/bin_debug.py lj > gpurun_out/final_bin_debug_l1.log 2>&1
  GRX_BIN_DEBUG=2 timeout 200 python tools/bin_debug.py lj > gpurun_out/final_bin_debug_l2.log 2>&1
  timeout 1200 python tests/tools/bench_all.py bfs_lj bfs_kron bfs_road sssp_road ssspu_road pr_kron bfs_twitter > gpurun_out/final_bench_all.log 2>&1
  bash tools/bench_generic.sh > gpurun_out/final_generic.log 2>&1
  timeout 300 python tests/tools/fuzz_gpu.py 150 > gpurun_out/final_fuzz.log 2>&1
fi
tail -2 gpurun_out/final_pytest_gpu.log; tail -1 gpurun_out/final_smoke.log; tail -2 gpurun_out/final_bench.log | cut -c1-300
