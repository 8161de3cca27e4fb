#!/bin/bash
# round 3, call 9: host-synthesis changes (inline Fr, recycled assignments): prover tests + bench
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_gpu_round3.py tests/test_cpp_api.py tests/test_gpu_r1cs.py tests/test_gpu_generator.py -m gpu -x -q > $OUT/tests.txt 2>&1; tail -2 $OUT/tests.txt
python - <<'PY' > $OUT/synth.txt 2>&1
from bellman_amd import _lib
lib=_lib.load()
for mode in (0,2,1,3):
    print("synthesis mode",mode,[round(lib.bh_test_synthesis_ms(1,(1<<20)-3,2020,mode),1) for _ in range(4)])
PY
cat $OUT/synth.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 200 $OUT/bench.json; tail -3 $OUT/bench.err
