#!/bin/bash
# round 4, call 25 (what is left of the budget): the -m gpu suite without the C5-scale file and the three slowest
# parameter-format cases, on the commit with the new host arithmetic (every multiexp result passes through host_fp.hpp's tail)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c25; mkdir -p $O
timeout 185 python -m pytest -q -x -m gpu tests --ignore=tests/test_gpu_scale.py -k "not 2_20 and not read_errors and not read_uncompressed_rules" --durations=5 > $O/tests.txt 2>&1
tail -9 $O/tests.txt
