#!/bin/bash
# Round 6, call 38: G2 window tables with 20-bit rows (13 rows, 2^19 buckets) against the 16-bit ones, 2^18 ... 2^20
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c38_g2_table_bits.txt
: > $out
timeout 900 python tools/profile_suite.py tsweep 2 18 20 16,18,19,20 >> $out 2>&1
timeout 900 python tools/profile_suite.py tsweep 2 19 20 20 0,16,32,64 >> $out 2>&1
