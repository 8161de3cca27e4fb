#!/bin/bash
# Round 6, call 35: 20-bit G1 window tables by default (2^19 ... 2^22 points): parity (the tests that touch table defaults),
# then bench.py with the new default against BELLMAN_HIP_TABLE_MAX_LOG2_G1=18 (the round's earlier behaviour), same box
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c35
mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_scale.py tests/test_gpu_boolean.py -q -x -m gpu 2>&1 | tail -5 > $out/tests.txt
timeout 600 python bench.py > $out/bench_new.json 2> $out/bench_new.err
BELLMAN_HIP_TABLE_MAX_LOG2_G1=18 timeout 600 python bench.py > $out/bench_old.json 2> $out/bench_old.err
timeout 600 python bench.py --no-cpu-baseline > $out/bench_new2.json 2> $out/bench_new2.err
