#!/bin/bash
# Round 6, call 34: window bits of G1 tables at 2^14 ... 2^19 re-swept with the two-stage sums (more buckets got cheaper)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c34_table_bits_mid.txt
: > $out
for r in 1 2; do
timeout 900 python tools/profile_suite.py tsweep 1 14 19 10,13,16,17,18,19,20 >> $out 2>&1
done
