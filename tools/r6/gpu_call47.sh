#!/bin/bash
# Round 6, call 47: kernel timeline of boolean-heavy G1 multiexps over the 20-bit table (2^20: 90 % booleans, 50 %, all ones)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r6c47_boolean_timeline.txt
: > $out
for mix in bool90 bool50 ones; do
  rm -rf /tmp/prof47
  (cd /tmp && MIX=$mix timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof47 -o p -- python $GRAFT_REPO_ROOT/tools/r6/boolean_mix.py 1 20 1 0 5 > /tmp/prof47.log 2>&1)
  f=$(find /tmp/prof47 -name '*kernel_trace.csv' | head -1)
  echo "## $mix" >> $out
  grep -v "^W\|^E" /tmp/prof47.log | tail -3 >> $out
  python tools/r6/trace_last_job.py $f >> $out 2>&1
done
