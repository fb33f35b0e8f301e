#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q -k "not full_size" > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c3_pytest.log)
timeout 300 python tools/ab_bfs.py lj bin > gpurun_out/c3_ab_lj.log 2>&1; echo "rc $?" >> gpurun_out/c3_ab_lj.log
timeout 300 python tools/ab_bfs.py kron bin > gpurun_out/c3_ab_kron.log 2>&1; echo "rc $?" >> gpurun_out/c3_ab_kron.log
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_fwd2; timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_fwd2 -o p -- python tools/run_algo.py bfs lj 3 0 merge_path forward > gpurun_out/c3_kt.log 2>&1
python - <<'PY' > gpurun_out/c3_dispatches.txt 2>&1
import csv,glob
f=glob.glob("gpurun_out/prof_fwd2/**/p_kernel_trace.csv",recursive=True)
rows=sorted(csv.DictReader(open(f[0])),key=lambda r:int(r["Start_Timestamp"]))
for r in rows[-40:]:
    n=r["Kernel_Name"]
    print("%-40s %9.1f us"%(n[:40],(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
tail -4 gpurun_out/c3_pytest.log; cut -c1-330 gpurun_out/c3_ab_lj.log; cut -c1-300 gpurun_out/c3_ab_kron.log; tail -24 gpurun_out/c3_dispatches.txt
