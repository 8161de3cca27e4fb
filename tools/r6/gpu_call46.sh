#!/bin/bash
# Round 6, call 46: G1 2^25 / 2^26 over window tables on request (20- / 22- / 24-bit rows: 13 / 12 / 11 rows) against the classic plan
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c46_g1_tables_2p25_2p26.txt
: > $out
timeout 1500 python tools/profile_suite.py tsweep 1 25 25 0,20,22,24 >> $out 2>&1
timeout 1500 python tools/profile_suite.py tsweep 1 26 26 0,24 >> $out 2>&1
