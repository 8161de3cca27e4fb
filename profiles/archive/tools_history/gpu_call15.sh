#!/bin/bash
# Round-2 GPU call 15: window-table bits 13 vs 16 at 2^12..2^17 (G2 and G1) with the current merge / reduction kernels
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c15
mkdir -p $OUT
export TMPDIR=/tmp
for l in 12 13 14 15 16 17; do BH_TABLE=1 python tools/tune_msm.py $l 13,16 0 2; done > $OUT/table_c_g2.txt 2>&1
for l in 12 13 14 15 16; do BH_TABLE=1 python tools/tune_msm.py $l 13,16 0 1; done > $OUT/table_c_g1.txt 2>&1
grep -h "log_n" $OUT/table_c_g2.txt $OUT/table_c_g1.txt
