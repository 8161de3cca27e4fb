#!/bin/bash
# Round 6, call 48: UPPER BOUND of what 30-bit-limb-resident Fp arithmetic could save in the G1 accumulation - a timing-only build
# (results are wrong by construction) whose Fp products neither slice their operands nor repack their results (BH_DIAG_NO_SLICE
# in msm_g1.hip only); same box, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c48_g1_noslice_upper_bound.txt
: > $out
for rep in 1 2 3; do
  echo "## shipped" >> $out
  timeout 200 python tools/profile_suite.py tsweep 1 20 20 0,20 >> $out 2>&1
  echo "## no slicing, no repack (timing only)" >> $out
  BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1 BELLMAN_HIP_LIB=$GRAFT_REPO_ROOT/bellman_amd/lib_noslice_g1/libbellman_hip.so timeout 200 python tools/profile_suite.py tsweep 1 20 20 0,20 >> $out 2>&1
done
