#!/bin/bash
# Round 6, call 32: kernel timeline of one multiexp (classic 2^20; 20-bit table K = 104) with the two-stage sums
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r6c32_timeline.txt
: > $out
for cfg in "sizes 1 20 20" "tsweep 1 20 20 20 104" "sizes 1 16 16" "sizes 1 18 18"; do
  rm -rf /tmp/prof32
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof32 -o p -- python tools/profile_suite.py $cfg > /tmp/prof32.log 2>&1
  f=$(find /tmp/prof32 -name '*kernel_trace.csv' | head -1)
  echo "## $cfg" >> $out
  grep "^G1" /tmp/prof32.log >> $out
  python tools/r6/trace_last_job.py $f >> $out 2>&1
done
