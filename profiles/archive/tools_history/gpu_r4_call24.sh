#!/bin/bash
# round 4, call 24 (the last 4 GPU-minutes): the host-side changes of the second session (mirror field arithmetic, evaluating
# combinations, 6x64 host Fp product, packed Variable, capture) on the GPU box: bench.py without the CPU baselines and the
# 2^24 proof, then the proof tests that compare C++ mirror, Python mirror and oracle
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c24; mkdir -p $O
timeout 170 python bench.py --no-cpu-baseline --c5-proof-log-n 0 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 600 $O/bench.err
timeout 100 python -m pytest -q -x -m gpu tests/test_gpu_groth16.py tests/test_cpp_api.py -k "not 2_20 and not concurrent" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 30 python tools/host_synthesis.py 20 5 > $O/host_synthesis.txt 2>&1; cat $O/host_synthesis.txt
