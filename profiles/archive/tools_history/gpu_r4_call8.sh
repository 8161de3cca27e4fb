#!/bin/bash
# Round 4, GPU call 8: FFT (lazily reduced butterflies, one-level tables), host-scalar multiexps in two halves, merge
# kernels at two wavefronts per SIMD - parity of what they touch, then timings.
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call8
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q > $OUT/round4.txt 2>&1; echo "round4: $(tail -1 $OUT/round4.txt)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/parity.txt 2>&1; echo "parity: $(tail -1 $OUT/parity.txt)"
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "fft" > $OUT/scale_fft.txt 2>&1; echo "scale fft: $(tail -1 $OUT/scale_fft.txt)"
timeout 600 python -m pytest tests/test_gpu_groth16.py -m gpu -x -q > $OUT/groth16.txt 2>&1; echo "groth16: $(tail -1 $OUT/groth16.txt)"
for ln in 16 20 22 24; do python tools/profile_suite.py fft $ln 10 >> $OUT/fft.txt 2>&1; done
BELLMAN_HIP_FFT_ONE_LEVEL=0 python tools/profile_suite.py fft 22 10 > $OUT/fft_two_level.txt 2>&1
cat $OUT/fft.txt; echo "two-level tables:"; cat $OUT/fft_two_level.txt
python tools/profile_suite.py sizes 2 10 20 > $OUT/g2.txt 2>&1
BH_SUITE_FLAGS=256 python tools/profile_suite.py sizes 2 10 17 > $OUT/g2_pairs.txt 2>&1
python tools/profile_suite.py sizes 1 14 20 > $OUT/g1.txt 2>&1
cat $OUT/g2.txt; echo "pairs forced:"; cat $OUT/g2_pairs.txt; cat $OUT/g1.txt
python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt
python tools/profile_suite.py proof 20 7 12 > $OUT/proof.txt 2>&1; grep create_proof $OUT/proof.txt
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-proof > $OUT/bench.json 2>/dev/null
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4_call8", "bench.json")).read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["config"]["device_ms"], "2 jobs", d["config"]["value_with_2_jobs_in_flight"], "incl upload", d["value_incl_scalar_upload"])
PY
