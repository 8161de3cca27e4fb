#!/bin/bash
# Round-2 GPU call 19: window-table bits 8 vs 13 for tiny vectors (2^8..2^11), the sizes of a MiMC-322 proof
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c19
mkdir -p $OUT
export TMPDIR=/tmp
for l in 8 9 10 11; do BH_TABLE=1 python tools/tune_msm.py $l 8,13 0 2; done > $OUT/table_c_g2.txt 2>&1
for l in 8 9 10 11; do BH_TABLE=1 python tools/tune_msm.py $l 8,13 0 1; done > $OUT/table_c_g1.txt 2>&1
grep -h "log_n" $OUT/table_c_g2.txt $OUT/table_c_g1.txt
