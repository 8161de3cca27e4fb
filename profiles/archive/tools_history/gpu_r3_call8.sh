#!/bin/bash
# round 3, call 8: lane-pair (K2) G1 reductions: parity, sizes A/B, MiMC
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c8
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "reductions_with or small_multiexp" > $OUT/k2.txt 2>&1; tail -3 $OUT/k2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_gpu_generator.py -m gpu -x -q > $OUT/parity.txt 2>&1; tail -3 $OUT/parity.txt
for f in 1 0; do
  echo "== BELLMAN_HIP_SUM_K2=$f" >> $OUT/k2_ab.txt
  BELLMAN_HIP_SUM_K2=$f python tools/profile_suite.py mimc 30 >> $OUT/k2_ab.txt 2>&1
  BELLMAN_HIP_SUM_K2=$f python tools/profile_suite.py sizes 1 10 20 >> $OUT/k2_ab.txt 2>&1
done
cat $OUT/k2_ab.txt
