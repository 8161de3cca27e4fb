#!/bin/bash
# Round 6, call 28: G1 window tables of 2^19 ... 2^22 points against the classic plan, with and without the touch of the next
# entry's record (msm_accumulate_kernel<.., TOUCH>; BELLMAN_HIP_ACC_TOUCH).  One process per setting.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c28_touch.txt
: > $out
for t in 0 1 0 1; do
  echo "## BELLMAN_HIP_ACC_TOUCH=$t" >> $out
  BELLMAN_HIP_ACC_TOUCH=$t timeout 600 python tools/profile_suite.py tsweep 1 19 22 0,16,20 >> $out 2>&1
done
echo "## parity of the table plan with the touch (default rule): boolean + scale tests at 2^20" >> $out
timeout 600 python -m pytest tests/test_gpu_boolean.py -q -x -m gpu -k "g1" 2>&1 | tail -3 >> $out
