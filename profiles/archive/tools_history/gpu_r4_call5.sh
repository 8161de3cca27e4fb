#!/bin/bash
# Round 4, GPU call 5: same-box A/B of the in-workgroup run folding (flags 512 = BH_MSM_NO_FOLD) over chunk lengths
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call5
mkdir -p $OUT
export TMPDIR=/tmp
python tools/profile_suite.py sweep 1 20 16 32,64,128 0,512 3 > $OUT/g1_20.txt 2>&1
python tools/profile_suite.py sweep 1 19 16 32,64 0,512 2 > $OUT/g1_19.txt 2>&1
python tools/profile_suite.py sweep 1 18 16 16,32 0,512 2 > $OUT/g1_18.txt 2>&1
python tools/profile_suite.py sweep 1 21 16 64,128,256 0,512 2 > $OUT/g1_21.txt 2>&1
python tools/profile_suite.py sweep 1 22 16 128,256,512 0,512 2 > $OUT/g1_22.txt 2>&1
python tools/profile_suite.py sweep 1 16 16 16,32 0,512 2 > $OUT/g1_16.txt 2>&1
python tools/profile_suite.py sweep 2 19 0 0 0,512 2 > $OUT/g2_19.txt 2>&1
python tools/profile_suite.py sweep 2 16 0 0 0,512,256,768 2 > $OUT/g2_16.txt 2>&1
cat $OUT/*.txt
