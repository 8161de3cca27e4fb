#!/bin/bash
# Round 6, call 53: sort tiles staged through LDS (sort_scatter_kernel<true>): parity, then A/B by environment switch
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c53_staged_scatter.txt
: > $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boolean.py tests/test_gpu_round4.py -q -x -m gpu 2>&1 | tail -3 >> $out
for e in 1 0 1 0; do
  echo "## BELLMAN_HIP_SORT_STAGED=$e" >> $out
  BELLMAN_HIP_SORT_STAGED=$e timeout 200 python tools/profile_suite.py sizes 1 16 20 >> $out 2>&1
  BELLMAN_HIP_SORT_STAGED=$e timeout 200 python tools/profile_suite.py sizes 1 22 24 >> $out 2>&1
  BELLMAN_HIP_SORT_STAGED=$e timeout 200 python tools/profile_suite.py sizes 2 18 20 >> $out 2>&1
done
BELLMAN_HIP_SORT_STAGED=1 timeout 200 python tools/profile_suite.py sizes 1 26 26 >> $out 2>&1
BELLMAN_HIP_SORT_STAGED=0 timeout 200 python tools/profile_suite.py sizes 1 26 26 >> $out 2>&1
