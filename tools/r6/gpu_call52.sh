#!/bin/bash
# Round 6, call 52: how much of a sort pass's scatter is its scattered 8-byte writes?  A timing-only build that writes every
# entry back to its own position (coalesced; results wrong by construction) against the shipped kernel: kernel statistics
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r6c52_scatter_writes.txt
: > $out
for v in shipped coalesced; do
  rm -rf /tmp/prof52
  if [ $v = coalesced ]; then export BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1 BELLMAN_HIP_LIB=$GRAFT_REPO_ROOT/bellman_amd/lib_coalesced/libbellman_hip.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof52 -o p -- python tools/profile_suite.py sizes 1 20 22 > /tmp/prof52.log 2>&1
  echo "## $v" >> $out
  grep "^G1" /tmp/prof52.log >> $out
  f=$(find /tmp/prof52 -name '*kernel_stats.csv' | head -1)
  grep "sort_scatter\|sort_hist\|msm_digits" $f | cut -d, -f1-4,6,7 >> $out
done
