#!/bin/bash
# Round 6, call 56: the small one-launch path declines plans whose sliver top row would overflow the bucket lists: tiny tables again
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c56_small_path_guard.txt
: > $out
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_groth16.py -q -x -m gpu 2>&1 | tail -2 >> $out
timeout 600 python tools/profile_suite.py tsweep 2 8 10 8,10,11,12,13 >> $out 2>&1
timeout 600 python tools/profile_suite.py tsweep 1 8 10 10,11,12,13 >> $out 2>&1
timeout 100 python tools/profile_suite.py mimc 30 >> $out 2>&1
