#!/bin/bash
# round 6, call 27: sweep with 512-thread workgroups, two per CU (GRX_SW_GEOM=512), item counts around it
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/ab_r6.py lj 20 "-" "GRX_SW_GEOM=512" "GRX_SW_GEOM=512,GRX_SW2_ITEMS=330" "GRX_SW_GEOM=512,GRX_SW2_ITEMS=600" "GRX_SW_GEOM=512,GRX_SW2_ITEMS=1000" "-" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_c27_sweep_geom.txt
cat gpurun_out/r6_c27_sweep_geom.txt
