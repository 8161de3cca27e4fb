#!/bin/bash
# Round-2 GPU call 5: parity after the plan-policy / single-copy changes, per-size tables, MiMC host timeline
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or bases or multiexp or fixed_base or k3" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
timeout 600 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_cpp_api.py -m gpu -q -x > $OUT/t_groth.txt 2>&1; tail -3 $OUT/t_groth.txt
python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1.txt 2>&1
python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2.txt 2>&1
python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt
BH_DEBUG=1 python tools/profile_suite.py mimc 1 > $OUT/mimc_trace.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_bench_smoke.py -m gpu -q -x > $OUT/t_bench_smoke.txt 2>&1; tail -5 $OUT/t_bench_smoke.txt
du -sh $OUT
