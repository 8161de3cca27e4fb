#!/bin/bash
# Round 6, call 39: G2 20- / 18-bit tables with the merge + reduction forced onto lane triples (flags 288 = LANE_TRIPLES | LANE_PAIRS)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c39_g2_table_bits_triples.txt
: > $out
BH_SUITE_FLAGS=288 timeout 900 python tools/profile_suite.py tsweep 2 19 20 16,18,20 >> $out 2>&1
BH_SUITE_FLAGS=288 timeout 900 python tools/profile_suite.py tsweep 2 20 20 20 26,32,52,64 >> $out 2>&1
