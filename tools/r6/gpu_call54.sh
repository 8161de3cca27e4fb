#!/bin/bash
# Round 6, call 54: window bits of tiny G2 tables (2^8 ... 2^12 points): 8-bit rows (the default) against 10 / 11 / 12 / 13 bits, which
# the one-launch small path (msm_small_fill_kernel) accepts; then the MiMC-322 proof
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c54_tiny_g2_bits.txt
: > $out
for r in 1 2; do timeout 600 python tools/profile_suite.py tsweep 2 8 12 8,10,11,12,13 >> $out 2>&1; done
