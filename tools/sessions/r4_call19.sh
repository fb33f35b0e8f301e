#!/bin/bash
# round 4, call 19: bins aimed at when the relax table is cut (GRX_RBIN_BINS, read when the handle builds its table: one process each)
mkdir -p gpurun_out
for g in lj kron; do
  for nb in 448 700 900 1000; do
    echo "== $g GRX_RBIN_BINS=$nb"
    GRX_RBIN_BINS=$nb timeout 300 python tools/ab_relax.py $g "" "" 2>&1 | grep -v amdgpu.ids | grep binned | cut -c1-100
  done
done > gpurun_out/r4c19_bins.log 2>&1; cat gpurun_out/r4c19_bins.log
