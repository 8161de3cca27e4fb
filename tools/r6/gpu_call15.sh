#!/bin/bash
# round 6, call 15: medium runs + big pieces in one launch (fused) against two launches, G1 and G2, same box, alternating
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c15
mkdir -p $OUT
for rep in 1 2; do
 for cfg in "2 10" "2 12" "2 14" "2 16" "1 14" "1 16" "1 17"; do
  set -- $cfg
  for f in 0 1; do
    echo "fused=$f $(BELLMAN_HIP_TAIL_FUSED=$f timeout 120 python tools/profile_suite.py msm $1 $2 12 | tail -1)"
  done
 done
done 2>&1 | tee $OUT/tail_split_ab.txt
for f in 0 1; do echo "fused=$f $(BELLMAN_HIP_TAIL_FUSED=$f timeout 120 python tools/profile_suite.py mimc 30 | tail -1)"; done 2>&1 | tee -a $OUT/tail_split_ab.txt
for f in 0 1; do echo "fused=$f bool50 G2 2^19: $(BELLMAN_HIP_TAIL_FUSED=$f MIX=bool50 timeout 120 python tools/r6/boolean_mix.py 2 19 1 0 9 | tail -1)"; done 2>&1 | tee -a $OUT/tail_split_ab.txt
