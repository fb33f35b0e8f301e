#!/bin/bash
# call 26: source level claims on the bitmap
mkdir -p gpurun_out
for g in lj kron twitter; do
  timeout 600 python tools/ab_bu.py $g 30 2>&1 | grep -v amdgpu.ids > gpurun_out/c26_ab_bu_$g.log; echo "ab $g rc $?"
  grep -v "^source" gpurun_out/c26_ab_bu_$g.log | cut -c1-110
done
