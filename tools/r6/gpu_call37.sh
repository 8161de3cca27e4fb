#!/bin/bash
# Round 6, call 37: rounds of the 20-bit table plan's accumulation (chunk length = entries / (chip lanes x rounds)) inside a
# 2^20-constraint proof and stand-alone
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c37_table_rounds.txt
: > $out
for rep in 1 2; do
for r in 1 2 3 4; do
  echo "## BELLMAN_HIP_TABLE_ROUNDS=$r" >> $out
  BELLMAN_HIP_TABLE_ROUNDS=$r timeout 100 python tools/profile_suite.py proof 20 9 1 2>&1 | grep create_proof >> $out
  BELLMAN_HIP_TABLE_ROUNDS=$r timeout 100 python tools/profile_suite.py proof 20 9 12 2>&1 | grep create_proof >> $out
  BELLMAN_HIP_TABLE_ROUNDS=$r timeout 100 python tools/profile_suite.py sizes 1 19 21 >> $out 2>&1
done
echo "## BELLMAN_HIP_TABLE_MAX_LOG2_G1=18" >> $out
BELLMAN_HIP_TABLE_MAX_LOG2_G1=18 timeout 100 python tools/profile_suite.py proof 20 9 1 2>&1 | grep create_proof >> $out
BELLMAN_HIP_TABLE_MAX_LOG2_G1=18 timeout 100 python tools/profile_suite.py proof 20 9 12 2>&1 | grep create_proof >> $out
done
