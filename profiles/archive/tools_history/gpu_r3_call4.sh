#!/bin/bash
# round 3, call 4: the accumulation chain A/B (one proof, twelve threads), timeline, parity, bench
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c4
mkdir -p $OUT
export TMPDIR=/tmp
for c in 1 0 1 0; do
  echo "== BELLMAN_HIP_ACC_CHAIN=$c" >> $OUT/chain.txt
  BELLMAN_HIP_ACC_CHAIN=$c python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof >> $OUT/chain.txt
done
cat $OUT/chain.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- python tools/profile_suite.py proof 20 3 1 > $OUT/trace.log 2>&1
grep create_proof $OUT/trace.log
f=$(ls $OUT/trace/*kernel_trace.csv | head -1); (head -1 $f; tail -3000 $f) > $OUT/proof_trace.csv; rm -rf $OUT/trace
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_gpu_round3.py -m gpu -x -q > $OUT/parity.txt 2>&1; tail -2 $OUT/parity.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 200 $OUT/bench.json; tail -3 $OUT/bench.err
