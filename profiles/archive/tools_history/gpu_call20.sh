#!/bin/bash
# Round-2 GPU call 20: parity + MiMC timing with 8-bit window tables for tiny G2 vectors
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c20
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or bases or multiexp or fixed_base or k3 or table" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
timeout 600 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_gpu_r1cs.py tests/test_gpu_generator.py -m gpu -q -x > $OUT/t_groth.txt 2>&1; tail -3 $OUT/t_groth.txt
python tools/profile_suite.py mimc 40 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt
python tools/profile_suite.py sizes 2 8 12 > $OUT/sizes_g2.txt 2>&1; cat $OUT/sizes_g2.txt
