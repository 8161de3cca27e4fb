#!/bin/bash
# Round 6, call 51: the counters of a sort pass scanned by one launch (scan_tile_last_kernel): parity, then A/B by environment switch
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c51_fused_scan.txt
: > $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boolean.py tests/test_gpu_round4.py tests/test_gpu_groth16.py -q -x -m gpu 2>&1 | tail -3 >> $out
for e in 1 0 1 0; do
  echo "## BELLMAN_HIP_SORT_FUSED_SCAN=$e" >> $out
  BELLMAN_HIP_SORT_FUSED_SCAN=$e timeout 200 python tools/profile_suite.py sizes 1 14 20 >> $out 2>&1
  BELLMAN_HIP_SORT_FUSED_SCAN=$e timeout 200 python tools/profile_suite.py sizes 2 16 18 >> $out 2>&1
done
