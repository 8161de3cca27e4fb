#!/bin/bash
# Round 4, GPU call 12: merge of the straddling runs - owner lane folds at most `walk` chunks, longer runs are queued for
# G lanes each: walk 4 / G 8 (shipped) against walk 1 with G 2, 4, 8 (one addition per owner lane, the 8 % of runs that span
# three chunks folded densely by the queue kernel)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call12
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
  for cfg in "4 8" "1 2" "1 4" "1 8" "2 2" "2 4"; do
    set -- $cfg
    echo "== walk $1 lanes $2" >> $OUT/g1_20.txt
    BELLMAN_HIP_MERGE_WALK=$1 BELLMAN_HIP_RUN_LANES=$2 python tools/profile_suite.py sweep 1 20 0 0 0 2 >> $OUT/g1_20.txt 2>&1
  done
done
cat $OUT/g1_20.txt
for cfg in "4 8" "1 2" "1 4"; do
  set -- $cfg
  echo "== walk $1 lanes $2" >> $OUT/other.txt
  BELLMAN_HIP_MERGE_WALK=$1 BELLMAN_HIP_RUN_LANES=$2 python tools/profile_suite.py sweep 1 18 0 0 0 2 >> $OUT/other.txt 2>&1
  BELLMAN_HIP_MERGE_WALK=$1 BELLMAN_HIP_RUN_LANES=$2 python tools/profile_suite.py sweep 1 22 0 0 0 1 >> $OUT/other.txt 2>&1
done
cat $OUT/other.txt
BELLMAN_HIP_MERGE_WALK=1 BELLMAN_HIP_RUN_LANES=2 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "msm or runs or reductions" > $OUT/parity.txt 2>&1; echo "parity (walk 1, 2 lanes): $(tail -1 $OUT/parity.txt)"
