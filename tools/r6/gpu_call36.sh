#!/bin/bash
# Round 6, call 36: kernel timeline of a 2^20-constraint proof with the 20-bit G1 tables against BELLMAN_HIP_TABLE_MAX_LOG2_G1=18
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in 22 18; do
  rm -rf /tmp/prof36
  BELLMAN_HIP_TABLE_MAX_LOG2_G1=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof36 -o p -- python tools/profile_suite.py proof 20 3 1 > gpurun_out/r6c36_proof_$v.log 2>&1
  f=$(find /tmp/prof36 -name '*kernel_trace.csv' | head -1)
  (head -1 $f; tail -3000 $f) > /tmp/proof_trace.csv
  python tools/proof_timeline.py /tmp/proof_trace.csv "G1 tables up to 2^$v" > gpurun_out/r6c36_proof_timeline_$v.txt 2>&1
  for i in 1 2; do BELLMAN_HIP_TABLE_MAX_LOG2_G1=$v timeout 100 python tools/profile_suite.py proof 20 7 1 2>&1 | grep create_proof >> gpurun_out/r6c36_proof_$v.txt; done
done
