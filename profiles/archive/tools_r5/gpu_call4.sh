#!/bin/bash
# round 5, call 4: the GLV form of the classic G1 plan - parity where it runs (2^19 ... 2^26 terms, density + skip, error
# semantics, shards, proofs) and same-process A/B against the classic plan (flag 512 = BH_MSM_NO_GLV), K = 32 / 64
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c4
mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 60 python tools/profile_suite.py sweep 1 20 0 0,32,64 0,512 2 > $OUT/sweep_2p20.txt 2>&1; cat $OUT/sweep_2p20.txt
echo "sweep done [$(( $(date +%s) - t0 )) s]"
timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py::test_large_host_scalar_multiexp_error_semantics \
  tests/test_gpu_round4.py::test_g1_window_table_at_128_byte_stride "tests/test_gpu_round3.py::test_sharded_multiexp_contexts_in_one_process" \
  tests/test_gpu_round3.py::test_sharded_multiexp_error_semantics tests/test_gpu_round3.py::test_call_sites_chain_circuit \
  tests/test_gpu_scale.py::test_msm_2_23_density_and_skip "tests/test_gpu_scale.py::test_msm_c5_scale_matches_oracle" tests/test_gpu_groth16.py \
  -x -q --durations=8 > $OUT/tests.txt 2>&1; tail -16 $OUT/tests.txt
echo "tests done [$(( $(date +%s) - t0 )) s]"
timeout 60 python tools/profile_suite.py sizes 1 19 24 > $OUT/sizes_glv.txt 2>&1
BH_SUITE_FLAGS=512 timeout 60 python tools/profile_suite.py sizes 1 19 24 > $OUT/sizes_classic.txt 2>&1
cat $OUT/sizes_glv.txt $OUT/sizes_classic.txt
timeout 150 python bench.py --no-cpu-baseline --c5-proof-log-n 0 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json; tail -2 $OUT/bench.err
echo "all done [$(( $(date +%s) - t0 )) s]"
