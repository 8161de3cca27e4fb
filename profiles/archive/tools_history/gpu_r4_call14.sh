#!/bin/bash
# Round 4, GPU call 14: windowed fixed-base multiplication (generate_parameters, fixture bases) - parity of everything that
# generates points with it, set-up times; G1 sizes with the window tables extended to 2^18
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call14
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_params_io.py -m gpu -x -q > $OUT/gen.txt 2>&1; echo "generator + params io: $(tail -1 $OUT/gen.txt)"
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "2-20 or 1-23" > $OUT/scale.txt 2>&1; echo "scale: $(tail -1 $OUT/scale.txt)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "msm or held or fixed or table" > $OUT/parity.txt 2>&1; echo "parity: $(tail -1 $OUT/parity.txt)"
python - <<'PY' > $OUT/setup.txt 2>&1
import sys, time
sys.path.insert(0, ".")
import bellman_amd
from bellman_amd import groth16 as pg
from bench import G1_GEN_MONT, G2_GEN_MONT
w = bellman_amd.Worker(0)
for log_n in (16, 20, 22):
    rounds = (1 << log_n) - 3
    t0 = time.perf_counter(); r1cs = pg.R1CS.from_demo(w, 1, rounds, 2020); cap = time.perf_counter() - t0
    for rep in range(2):
        t0 = time.perf_counter()
        params = pg.Parameters.generate(w, r1cs, G1_GEN_MONT, G2_GEN_MONT, alpha=48577, beta=22580, gamma=53332, delta=5481, tau=3673)
        gen = time.perf_counter() - t0
        t0 = time.perf_counter(); blob = params.write(); wr = time.perf_counter() - t0
        print("2^%d constraints: capture %.0f ms, generate_parameters %.0f ms, Parameters::write %.0f ms (%.0f MB)" % (log_n, cap * 1e3, gen * 1e3, wr * 1e3, len(blob) / 1e6), flush=True)
        params.release(); del blob
    r1cs.release(); w.trim()
PY
cat $OUT/setup.txt
python tools/profile_suite.py sizes 1 16 19 > $OUT/g1.txt 2>&1; cat $OUT/g1.txt
