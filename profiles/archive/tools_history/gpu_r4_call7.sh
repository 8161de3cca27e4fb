#!/bin/bash
# Round 4, GPU call 7: the whole -m gpu suite after the library split (product / test library), the new asynchronous
# entry points, lane pairs as the default of large G2 accumulations, fused Y3; then bench.py.
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call7
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/gputests.txt 2>&1; tail -18 $OUT/gputests.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
