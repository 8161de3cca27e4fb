#!/bin/bash
# Round 6, call 44: G1 2^23 / 2^24 with a 20-bit window table (14 / 28 GB) against the classic plans
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c44_g1_tables_2p23_2p24.txt
: > $out
timeout 900 python tools/profile_suite.py tsweep 1 23 24 0,20 >> $out 2>&1
