#!/bin/bash
# round 3, call 5: h block in front of the accumulation chain; timeline; round-3 tests; bench
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c5
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2; do python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof >> $OUT/proof.txt; done
BELLMAN_HIP_H_PRIORITY=1 python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof >> $OUT/proof.txt
cat $OUT/proof.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- python tools/profile_suite.py proof 20 3 1 > $OUT/trace.log 2>&1
grep create_proof $OUT/trace.log
f=$(ls $OUT/trace/*kernel_trace.csv | head -1); (head -1 $f; tail -3000 $f) > $OUT/proof_trace.csv; rm -rf $OUT/trace
timeout 600 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_gpu_round3.py -m gpu -x -q > $OUT/parity.txt 2>&1; tail -2 $OUT/parity.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 200 $OUT/bench.json; tail -3 $OUT/bench.err
