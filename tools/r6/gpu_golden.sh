#!/bin/bash
# Regenerates tests/golden/scale_oracle.json: runs the large seeded parity cases with the CPU oracle in the loop
# (BELLMAN_GOLDEN_REGEN=1, tests/golden_cache.py) on the GPU box - the inputs are generated on the device - and leaves the
# records in gpurun_out/golden/scale_oracle.json; copy that file to tests/golden/ afterwards.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/golden
cp tests/golden/scale_oracle.json gpurun_out/golden/scale_oracle.json   # (records of cases not re-run are kept)
BELLMAN_GOLDEN_REGEN=1 timeout 1500 python -m pytest tests/test_gpu_boolean.py tests/test_gpu_scale.py tests/test_gpu_groth16.py -m gpu -x -q -s --durations=12 \
  -k "boolean or c5_scale or 2_23_density or proof_2_22 or proof_2_24 or chain_2_20 or repeated" > gpurun_out/golden/regen.log 2>&1
tail -40 gpurun_out/golden/regen.log
ls -la gpurun_out/golden
