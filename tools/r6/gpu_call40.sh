#!/bin/bash
# Round 6, call 40: G2 tables of 16 / 18 / 20-bit rows with the reduction of big table sets on lane triples + two-stage sums
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c40_g2_tables_two_stage.txt
: > $out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "g2 or G2 or group" 2>&1 | tail -3 >> $out
for e in 1 0; do
echo "## BELLMAN_HIP_SUM_TWO_STAGE=$e" >> $out
BELLMAN_HIP_SUM_TWO_STAGE=$e timeout 900 python tools/profile_suite.py tsweep 2 16 21 16,18,20 >> $out 2>&1
done
echo "## classic plan, reductions on lane triples (flags 288) against the default" >> $out
BH_SUITE_FLAGS=288 timeout 300 python tools/profile_suite.py tsweep 2 20 20 0 >> $out 2>&1
timeout 300 python tools/profile_suite.py tsweep 2 20 20 0 >> $out 2>&1
