#!/bin/bash
# round 3, call 7: single-launch path for small multiexps: parity, sizes, MiMC
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c7
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "small_multiexp or held or call_sites_mimc" > $OUT/small.txt 2>&1; tail -3 $OUT/small.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_gpu_generator.py tests/test_cpp_api.py -m gpu -x -q > $OUT/parity.txt 2>&1; tail -3 $OUT/parity.txt
for f in 1 0; do
  echo "== BELLMAN_HIP_SMALL_FUSED=$f" >> $OUT/small_ab.txt
  BELLMAN_HIP_SMALL_FUSED=$f python tools/profile_suite.py mimc 30 >> $OUT/small_ab.txt 2>&1
  BELLMAN_HIP_SMALL_FUSED=$f python tools/profile_suite.py sizes 1 8 11 >> $OUT/small_ab.txt 2>&1
done
cat $OUT/small_ab.txt
