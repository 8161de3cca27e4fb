#!/bin/bash
# Round 6, call 45: one bucket set - the first sort pass drops the zero digits (BELLMAN_HIP_SORT_DROP_ZEROS): parity, then the scalar
# mixes with and without, and a boolean-heavy proof
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c45_drop_zeros.txt
: > $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boolean.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_groth16.py -q -x -m gpu 2>&1 | tail -4 >> $out
for d in 1 0 1 0; do
  echo "## BELLMAN_HIP_SORT_DROP_ZEROS=$d" >> $out
  for a in "1 20 1 0" "1 16 1 0" "2 20 1 0" "2 19 1 1"; do BELLMAN_HIP_SORT_DROP_ZEROS=$d timeout 200 python tools/r6/boolean_mix.py $a 9 >> $out 2>&1; done
  BELLMAN_HIP_SORT_DROP_ZEROS=$d timeout 100 python tools/profile_suite.py mimc 30 >> $out 2>&1
done
