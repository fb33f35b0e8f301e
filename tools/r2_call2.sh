#!/bin/bash
# per-dispatch timing + counters of the forward BFS kernels (binned levels)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/profile.sh fwd python tools/run_algo.py bfs lj 4 0 merge_path forward > gpurun_out/c2_profile.log 2>&1
python - <<'PY' > gpurun_out/c2_dispatches.txt 2>&1
import csv,glob,collections
f=glob.glob("gpurun_out/prof_fwd/kt/**/p_kernel_trace.csv",recursive=True)
rows=sorted(csv.DictReader(open(f[0])),key=lambda r:int(r["Start_Timestamp"]))
for r in rows:
    n=r["Kernel_Name"]
    if "bfs_" in n or "bin_" in n:
        print("%-40s %9.1f us grid %s wg %s lds %s vgpr %s sgpr %s"%(n[:40],(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r.get("Grid_Size"),r.get("Workgroup_Size"),r.get("LDS_Block_Size"),r.get("VGPR_Count"),r.get("SGPR_Count")))
PY
tail -60 gpurun_out/c2_dispatches.txt
grep -A40 "bfs_level_bin\|bfs_claim" gpurun_out/prof_fwd/summary.md | head -120
