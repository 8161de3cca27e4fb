#!/bin/bash
# Round 6, call 43: G2 20-bit tables from 2^20 points by default: parity, sizes, proofs (2^20 constraints) against 16-bit rows
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c43_g2_table20.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boolean.py tests/test_gpu_scale.py -q -x -m gpu -k "not g1" 2>&1 | tail -3 >> $out
for v in 20 99 20 99; do
  echo "## BELLMAN_HIP_G2_TABLE20_FROM=$v" >> $out
  BELLMAN_HIP_G2_TABLE20_FROM=$v timeout 100 python tools/profile_suite.py proof 20 9 1 2>&1 | grep create_proof >> $out
  BELLMAN_HIP_G2_TABLE20_FROM=$v timeout 100 python tools/profile_suite.py proof 20 9 12 2>&1 | grep create_proof >> $out
  BELLMAN_HIP_G2_TABLE20_FROM=$v timeout 100 python tools/profile_suite.py sizes 2 19 21 >> $out 2>&1
done
