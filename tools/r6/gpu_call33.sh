#!/bin/bash
# Round 6, call 33: classic plan against 16- and 20-bit G1 window tables at 2^19 ... 2^22 with the two-stage sums, the
# four-wavefront wide sums and one-round chunks for many-bucket tables
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c33_tables_after_sums.txt
: > $out
for r in 0 1 0 1; do
echo "## BELLMAN_HIP_TABLE_ONE_ROUND=$r" >> $out
BELLMAN_HIP_TABLE_ONE_ROUND=$r timeout 600 python tools/profile_suite.py tsweep 1 19 22 0,16,20 >> $out 2>&1
done
