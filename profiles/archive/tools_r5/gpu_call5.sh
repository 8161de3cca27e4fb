#!/bin/bash
# round 5, call 5: chunk length K against 2 K where the plan's rule was last swept with round-1 kernels
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c5
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "1 19 0,32,64" "1 20 0,32,64,128" "1 21 0,64,128" "1 22 0,128,256" "1 23 0,256,512" "2 19 0,64,128" "2 20 0,64,128"; do
  set -- $cfg
  timeout 90 python tools/profile_suite.py sweep $1 $2 0 $3 0 2 >> $OUT/k_sweep.txt 2>&1
done
cat $OUT/k_sweep.txt
