#!/bin/bash
# Round-2 GPU call 22: small proofs issue their assignment multiexps from a helper thread
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c22
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_gpu_r1cs.py tests/test_gpu_generator.py tests/test_cpp_api.py -m gpu -q -x > $OUT/t_groth.txt 2>&1; tail -3 $OUT/t_groth.txt
python tools/profile_suite.py mimc 40 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt
python tools/profile_suite.py mimc 40 >> $OUT/mimc.txt 2>&1; tail -1 $OUT/mimc.txt
