#!/bin/bash
# Round 4, GPU call 10: software-pipelined loads in the G1 accumulate kernel (next base point + the entry after next
# loaded right before the inline fused tail of the mixed addition, as the one-lane G2 kernel does): experimental build
# lib_exp_pipe (registers held to 256 by the launch bound, 55 spills) beside the shipped one; flags 2 = accumulator in
# LDS (190 VGPRs, no spills in the experimental build)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call10
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$GRAFT_REPO_ROOT/bellman_amd/lib_exp_pipe/libbellman_hip.so
for rep in 1 2; do
  python tools/profile_suite.py sweep 1 20 0 0 0,2 2 > $OUT/base_$rep.txt 2>&1
  BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1 BELLMAN_HIP_LIB=$EXP python tools/profile_suite.py sweep 1 20 0 0 0,2 2 > $OUT/pipe_$rep.txt 2>&1
done
for f in base_1 pipe_1 base_2 pipe_2; do echo "== $f"; cat $OUT/$f.txt; done
BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1 BELLMAN_HIP_LIB=$EXP timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $OUT/parity_pipe.txt 2>&1; echo "parity (pipelined build): $(tail -1 $OUT/parity_pipe.txt)"
