#!/bin/bash
# Round-2 GPU call 31: classic plan, c = 13 vs 16 at 2^17..2^20 (G1) with the current merge / reduction kernels
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c31
mkdir -p $OUT
export TMPDIR=/tmp
for l in 17 18 19 20; do BH_FLAGS=4 python tools/tune_msm.py $l 13,16 0 1; done > $OUT/classic_c_g1.txt 2>&1
cat $OUT/classic_c_g1.txt
