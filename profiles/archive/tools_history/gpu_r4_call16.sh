#!/bin/bash
# round 4, call 16: window tables for G1 vectors of 2^19 / 2^20 points - dense or at a 128-byte record stride - against the
# classic plan (which carries a 0.27 ms host tail of 256 doublings + 256 additions at 2^20); results first, then times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c16; mkdir -p $O
export BELLMAN_HIP_TABLE_MAX_LOG2=18
( timeout 300 python tools/table_pad_check.py 19 ) > $O/check.txt 2>&1; echo "check rc=$?" >> $O/check.txt
( BELLMAN_HIP_TABLE_PAD=0 timeout 200 python tools/table_pad_check.py 19 ) > $O/check_dense.txt 2>&1; echo "check rc=$?" >> $O/check_dense.txt
( timeout 400 python tools/profile_suite.py tsweep 1 19 20 0,16,19,20 ) > $O/tsweep_padded.txt 2>&1
( BELLMAN_HIP_TABLE_PAD=0 timeout 400 python tools/profile_suite.py tsweep 1 19 20 16,20 ) > $O/tsweep_dense.txt 2>&1
( timeout 400 python tools/profile_suite.py tsweep 1 21 22 0,16,20 ) > $O/tsweep_padded_21_22.txt 2>&1
tail -3 $O/check.txt $O/check_dense.txt; cat $O/tsweep_padded.txt $O/tsweep_dense.txt $O/tsweep_padded_21_22.txt | grep -v amdgpu.ids
