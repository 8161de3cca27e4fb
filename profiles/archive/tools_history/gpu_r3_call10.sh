#!/bin/bash
# round 3, call 10: capture through the sink hook - r1cs / generator / prover tests, capture time
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c10
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_r1cs.py tests/test_gpu_generator.py tests/test_gpu_groth16.py tests/test_cpp_api.py tests/test_gpu_proof_sharded.py -m gpu -x -q > $OUT/tests.txt 2>&1; tail -2 $OUT/tests.txt
python - <<'PY' > $OUT/capture.txt 2>&1
import time, bellman_amd
from bellman_amd import groth16 as pg
w = bellman_amd.Worker(0)
for lg in (20, 22):
    t0 = time.perf_counter(); r = pg.R1CS.from_demo(w, 1, (1 << lg) - 3, 2020); t1 = time.perf_counter()
    print("R1CS capture 2^%d: %.0f ms" % (lg, (t1 - t0) * 1e3)); r.release()
PY
cat $OUT/capture.txt
