#!/bin/bash
# Round 6, call 30: kernel statistics of the 20-bit G1 window table plan at 2^20 (reduce phase breakdown), K = plan's and 104
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for kk in 0 104; do
  rm -rf /tmp/prof30
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof30 -o p -- python tools/profile_suite.py tsweep 1 20 20 20 $kk > gpurun_out/r6c30_K${kk}.log 2>&1
  f=$(find /tmp/prof30 -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r6c30_table20_K${kk}_kernel_stats.csv
done
