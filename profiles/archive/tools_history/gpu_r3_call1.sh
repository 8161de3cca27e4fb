#!/bin/bash
# round 3, call 1: the new round-3 parity tests, the 2^24 proof + FFT 2^26 scale tests, then bench.py (new legs)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c1
mkdir -p $OUT
export TMPDIR=/tmp
free -g > $OUT/mem.txt; nproc >> $OUT/mem.txt
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s > $OUT/round3.txt 2>&1; tail -5 $OUT/round3.txt
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -s -k "proof_2_24 or fft_above and 26 or msm_c5 and 2-2" > $OUT/scale.txt 2>&1; tail -5 $OUT/scale.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; tail -5 $OUT/bench.err
