#!/bin/bash
# Round-2 GPU call 7: parity with the multi-wavefront K3 reduction; G2 per-size table
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c7
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or bases or multiexp or fixed_base or k3" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
timeout 600 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py -m gpu -q -x > $OUT/t_groth.txt 2>&1; tail -3 $OUT/t_groth.txt
python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2.txt 2>&1; cat $OUT/sizes_g2.txt
python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt
