#!/bin/bash
# Round-2 GPU call 16: parity + per-size tables after the table-bits change
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c16
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or bases or multiexp or fixed_base or k3" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
timeout 600 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py -m gpu -q -x > $OUT/t_groth.txt 2>&1; tail -3 $OUT/t_groth.txt
python tools/profile_suite.py sizes 1 13 17 > $OUT/sizes_g1.txt 2>&1; cat $OUT/sizes_g1.txt
python tools/profile_suite.py sizes 2 13 17 > $OUT/sizes_g2.txt 2>&1; cat $OUT/sizes_g2.txt
