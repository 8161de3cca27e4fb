#!/bin/bash
# round 5, call 6: soak (incl. the FFT table cache under a tight budget from four threads) and a second bench run for the
# box-to-box spread of the final numbers
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c6
mkdir -p $OUT
export TMPDIR=/tmp
timeout 240 python tools/soak.py > $OUT/soak.txt 2>&1; tail -6 $OUT/soak.txt | cut -c1-400
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
