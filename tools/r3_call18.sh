#!/bin/bash
# call 18: kernel timeline (start / duration / gap) of one direction-optimising and one forward search, unprofiled async runs
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl_do
timeout 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_do -o p -- python bench.py --only bfs --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/c18_tl.log 2>&1
python tools/timeline.py gpurun_out/tl_do 8 "bfs_level_kernel<" > gpurun_out/c18_timeline_do.txt 2>&1
python tools/timeline.py gpurun_out/tl_do 9 "bfs_level_kernel<" >> gpurun_out/c18_timeline_do.txt 2>&1
python tools/timeline.py gpurun_out/tl_do 8 "bfs_level_bin_kernel" >> gpurun_out/c18_timeline_do.txt 2>&1
rm -rf gpurun_out/tl_do
cat gpurun_out/c18_timeline_do.txt
