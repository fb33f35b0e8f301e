#!/bin/bash
# round 5, call 15: kernel sequence of one weighted SSSP on the LJ stand-in, one direction-optimising BFS, one forward BFS on the deep
# stand-in; binning threshold on the deep stand-in
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
bash tools/kt_last.sh sssp_w_lj fill_f32 -- python tools/sssp_loop.py lj 5
bash tools/kt_last.sh do_lj reset_seed -- python tools/fwd_loop.py lj 10 do
bash tools/kt_last.sh fwd_deep reset_seed -- python tools/fwd_loop.py deep 10 fwd
GRX_BIN_MIN_EDGES=2097152 bash tools/kt_last.sh fwd_deep_bin2M reset_seed -- python tools/fwd_loop.py deep 10 fwd
bash tools/kt_last.sh do_deep reset_seed -- python tools/fwd_loop.py deep 10 do
} > gpurun_out/r5c15_kt.log 2>&1
cut -c1-2500 gpurun_out/r5c15_kt.log
