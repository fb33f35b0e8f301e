#!/bin/bash
# round 6, call 19: the two new partition tests (one rank == the engine at full size; eight slices shrink a rank's work)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_distributed.py -m gpu -q -x --durations=6 -k "one_rank_partition_is_the_engine or eight_slices_shrink" > gpurun_out/r6_c19_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c19_pytest.log)
tail -25 gpurun_out/r6_c19_pytest.log; cat gpurun_out/part_sim_c5_1_vs_8.json
