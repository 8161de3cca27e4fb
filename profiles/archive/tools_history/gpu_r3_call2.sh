#!/bin/bash
# round 3, call 2: effective chunk length (sparse density), async proofs; A/B: G1 LDS accumulator (3 wavefronts), G1 window table at 2^20,
# half-dense shapes; proof timeline; bench
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_gpu_r1cs.py -m gpu -x -q > $OUT/parity.txt 2>&1; tail -3 $OUT/parity.txt
timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "async or call_sites" > $OUT/round3_subset.txt 2>&1; tail -3 $OUT/round3_subset.txt
python tools/r3_ab.py g1flags > $OUT/ab_g1flags.txt 2>&1; cat $OUT/ab_g1flags.txt
python tools/r3_ab.py halfdense 1 > $OUT/ab_halfdense.txt 2>&1; python tools/r3_ab.py halfdense 2 >> $OUT/ab_halfdense.txt 2>&1; cat $OUT/ab_halfdense.txt
python tools/r3_ab.py g1table > $OUT/ab_g1table.txt 2>&1; cat $OUT/ab_g1table.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- python tools/profile_suite.py proof 20 3 1 > $OUT/trace.log 2>&1
grep create_proof $OUT/trace.log
f=$(ls $OUT/trace/*kernel_trace.csv | head -1); (head -1 $f; tail -3000 $f) > $OUT/proof_trace.csv; rm -rf $OUT/trace
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json; tail -3 $OUT/bench.err
