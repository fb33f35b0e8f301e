#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c11_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c11_pytest.log)
tail -4 gpurun_out/c11_pytest.log
timeout 300 python tools/ab_bfs.py lj bin2 > gpurun_out/c11_ab_lj.log 2>&1; grep "TD " gpurun_out/c11_ab_lj.log | cut -c1-330
timeout 200 python tools/run_algo.py bfs road 3 0 merge_path forward > gpurun_out/c11_bfs_road.log 2>&1; tail -1 gpurun_out/c11_bfs_road.log | cut -c1-200
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c11_smoke.log 2>&1; tail -1 gpurun_out/c11_smoke.log
