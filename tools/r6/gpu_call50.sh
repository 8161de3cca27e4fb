#!/bin/bash
# Round 6, call 50: the mid-size table-plan fuzz (30 extra seeds through an env range), the random-shape fuzz with 40 extra
# seeds, and a 100-s soak of boolean-heavy multiexps from four host threads on the new default plans
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c50_fuzz_soak.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_boolean.py -q -x -m gpu -k "fuzz_mid" 2>&1 | tail -4 >> $out
BH_FUZZ_EXTRA=40 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fuzz" 2>&1 | tail -3 >> $out
timeout 400 python tools/r6/soak_long_runs.py 100 4 >> $out 2>&1
