#!/bin/bash
# Round 6, call 41: kernel timeline of one G2 multiexp over a 20-bit table (2^20, 2^19)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r6c41_g2_timeline.txt
: > $out
for cfg in "tsweep 2 20 20 20" "tsweep 2 19 19 20" "tsweep 2 20 20 16"; do
  rm -rf /tmp/prof41
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof41 -o p -- python tools/profile_suite.py $cfg > /tmp/prof41.log 2>&1
  f=$(find /tmp/prof41 -name '*kernel_trace.csv' | head -1)
  echo "## $cfg" >> $out
  grep "^G2" /tmp/prof41.log >> $out
  python tools/r6/trace_last_job.py $f | grep -v "scan_\|sort_" >> $out 2>&1
done
