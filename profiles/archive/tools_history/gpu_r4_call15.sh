#!/bin/bash
# Round 4, GPU call 15: chunk length K of the G2 table plans now that the accumulation runs on lane pairs (the plan sizes
# the launch for one wavefront per SIMD of 64 LANES; a pair wavefront carries 32 workers at two wavefronts per SIMD);
# cheaper identity test of affine records (parity subset + G1 timing)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call15
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $OUT/parity.txt 2>&1; echo "parity: $(tail -1 $OUT/parity.txt)"
python tools/profile_suite.py sweep 2 19 0 0,16,32,48,64,96,128 0 2 > $OUT/g2_19.txt 2>&1; cat $OUT/g2_19.txt
python tools/profile_suite.py sweep 2 20 0 0,32,64,128 0 2 > $OUT/g2_20.txt 2>&1; cat $OUT/g2_20.txt
python tools/profile_suite.py sweep 2 16 0 0,8,16,32 0 2 > $OUT/g2_16.txt 2>&1; cat $OUT/g2_16.txt
python tools/profile_suite.py sweep 2 17 0 0,8,16,32 0 2 > $OUT/g2_17.txt 2>&1; cat $OUT/g2_17.txt
python tools/profile_suite.py sweep 1 20 0 0 0 3 > $OUT/g1_20.txt 2>&1; cat $OUT/g1_20.txt
