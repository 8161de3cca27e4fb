#!/bin/bash
# round 3, call 3: table-plan oversubscription sweep (standalone G2 + inside a proof), h block first / high priority A/B, timeline
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c3
mkdir -p $OUT
export TMPDIR=/tmp
for o in 1 2 4 8; do
  echo "== BELLMAN_HIP_TABLE_OVERSUB=$o" >> $OUT/oversub.txt
  BELLMAN_HIP_TABLE_OVERSUB=$o python tools/r3_ab.py halfdense 2 >> $OUT/oversub.txt 2>&1
  BELLMAN_HIP_TABLE_OVERSUB=$o python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof >> $OUT/oversub.txt
done
echo "== H_PRIORITY=0 (oversub 4)" >> $OUT/oversub.txt
BELLMAN_HIP_H_PRIORITY=0 python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof >> $OUT/oversub.txt
cat $OUT/oversub.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py -m gpu -x -q > $OUT/parity.txt 2>&1; tail -2 $OUT/parity.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- python tools/profile_suite.py proof 20 3 1 > $OUT/trace.log 2>&1
grep create_proof $OUT/trace.log
f=$(ls $OUT/trace/*kernel_trace.csv | head -1); (head -1 $f; tail -3000 $f) > $OUT/proof_trace.csv; rm -rf $OUT/trace
