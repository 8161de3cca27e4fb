#!/bin/bash
# round 3, call 6: two-phase issue (held jobs) in create_proof: A/B, timeline, tests
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c6
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py -m gpu -x -q > $OUT/parity.txt 2>&1; tail -2 $OUT/parity.txt
for h in 1 0 1 0; do
  echo "== BELLMAN_HIP_PROOF_HOLD=$h" >> $OUT/hold.txt
  BELLMAN_HIP_PROOF_HOLD=$h python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof >> $OUT/hold.txt
done
cat $OUT/hold.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- python tools/profile_suite.py proof 20 3 1 > $OUT/trace.log 2>&1
grep create_proof $OUT/trace.log
f=$(ls $OUT/trace/*kernel_trace.csv | head -1); (head -1 $f; tail -3000 $f) > $OUT/proof_trace.csv; rm -rf $OUT/trace
