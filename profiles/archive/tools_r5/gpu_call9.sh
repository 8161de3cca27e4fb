#!/bin/bash
# round 5, call 9: chunk length of the window-table plans (G1 2^12 ... 2^18, G2 2^12 ... 2^18) against multiples of the plan's choice
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c9
mkdir -p $OUT
export TMPDIR=/tmp
for lg in 12 14 15 16 17 18; do
  timeout 60 python tools/profile_suite.py sweep 1 $lg 0 0,8,16,32,64,128 0 1 >> $OUT/k_sweep_tables.txt 2>&1
done
for lg in 12 14 16 17 18; do
  timeout 60 python tools/profile_suite.py sweep 2 $lg 0 0,8,16,32,64,128,256 0 1 >> $OUT/k_sweep_tables.txt 2>&1
done
cat $OUT/k_sweep_tables.txt
