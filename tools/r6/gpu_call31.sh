#!/bin/bash
# Round 6, call 31: bit sums over the selected half only, wide (multi-wavefront) lane-pair sums, two-stage row / column sums:
# parity, then same-box A/B by environment switch (one process per setting).
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c31_sums.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boolean.py -q -x -m gpu 2>&1 | tail -4 >> $out
run() {  # label, env...
  echo "## $1" >> $out; shift
  env "$@" timeout 300 python tools/profile_suite.py sizes 1 19 20 >> $out 2>&1
  env "$@" timeout 300 python tools/profile_suite.py tsweep 1 20 20 20 104 >> $out 2>&1
  env "$@" timeout 300 python tools/profile_suite.py sizes 1 16 18 >> $out 2>&1
}
run "baseline (no wide, one stage)" BELLMAN_HIP_SUM_TWO_STAGE=0 BELLMAN_HIP_SUM_WIDE=0
run "wide only" BELLMAN_HIP_SUM_TWO_STAGE=0
run "two-stage lane pairs len 16 (default)" BELLMAN_HIP_SUM_TWO_STAGE=1
run "two-stage lane pairs len 8" BELLMAN_HIP_SUM_TWO_STAGE=1 BELLMAN_HIP_SUM_TWO_LEN=8
run "two-stage lane pairs len 32" BELLMAN_HIP_SUM_TWO_STAGE=1 BELLMAN_HIP_SUM_TWO_LEN=32
run "two-stage one lane len 8" BELLMAN_HIP_SUM_TWO_STAGE=2 BELLMAN_HIP_SUM_TWO_LEN=8
run "two-stage one lane len 16" BELLMAN_HIP_SUM_TWO_STAGE=2 BELLMAN_HIP_SUM_TWO_LEN=16
run "baseline again" BELLMAN_HIP_SUM_TWO_STAGE=0 BELLMAN_HIP_SUM_WIDE=0
run "default again" A=1
