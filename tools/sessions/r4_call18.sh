#!/bin/bash
# round 4, call 18: emission of the sweeps with 6 passes of row-offset loads in flight instead of 3 (library variant libgrx_g6.so), alternating processes
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in "" gunrock_amd/libgrx_g6.so; do
    echo "== lib ${lib:-default}"
    GRX_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python tools/ab_r4.py lj 20 2>&1 | grep -v amdgpu.ids | grep 'four sub-counters per bin\|four sub-counters again' | cut -c1-330
    GRX_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python tools/ab_relax.py lj "" 2>&1 | grep -v amdgpu.ids | grep 'binned' | cut -c1-120
  done
done > gpurun_out/r4c18_emit_g.log 2>&1; cat gpurun_out/r4c18_emit_g.log
