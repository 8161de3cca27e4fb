#!/bin/bash
# Round 6, call 49: kernel timeline of G1 multiexps over 16-bit tables at 2^16 and 2^18 (where does their 0.54-ms reduce go?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r6c49_table16_mid_timeline.txt
: > $out
for cfg in "tsweep 1 16 16 16" "tsweep 1 18 18 16" "tsweep 1 18 18 13"; do
  rm -rf /tmp/prof49
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof49 -o p -- python tools/profile_suite.py $cfg > /tmp/prof49.log 2>&1
  f=$(find /tmp/prof49 -name '*kernel_trace.csv' | head -1)
  echo "## $cfg" >> $out
  grep "^G1" /tmp/prof49.log >> $out
  python tools/r6/trace_last_job.py $f | grep -v "scan_" >> $out 2>&1
done
