#!/bin/bash
# Round-2 GPU call 14: one out-of-line Fp2 product with the second operand through an LDS staging slot
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c14
mkdir -p $OUT
export TMPDIR=/tmp
./tools/_build/mb_occ > $OUT/mb_occ.txt 2>&1; grep "waves/SIMD [12] .*Fp2" $OUT/mb_occ.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or bases or multiexp or fixed_base or k3 or field or point" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2.txt 2>&1; cat $OUT/sizes_g2.txt
