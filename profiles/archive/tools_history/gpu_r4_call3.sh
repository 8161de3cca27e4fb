#!/bin/bash
# Round 4, GPU call 3: runs folded inside the accumulation workgroup (msm_ec.cuh FOLD) - parity, then stage times;
# small G2 sizes with lane pairs vs the default (lane triples below 2^18) to set the switch-over.
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q > $OUT/parity.txt 2>&1
echo "parity: $(tail -1 $OUT/parity.txt)"
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "msm" > $OUT/scale.txt 2>&1
echo "scale: $(tail -1 $OUT/scale.txt)"
for rep in 1 2; do
  python tools/profile_suite.py sizes 1 14 20 > $OUT/g1_$rep.txt 2>&1
  python tools/profile_suite.py sizes 2 10 20 > $OUT/g2_$rep.txt 2>&1
  BH_SUITE_FLAGS=256 python tools/profile_suite.py sizes 2 10 17 > $OUT/g2_pairs_$rep.txt 2>&1
done
for f in $OUT/g1_*.txt $OUT/g2_*.txt; do echo "== $(basename $f)"; cat $f; done
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only > $OUT/bench.json 2>/dev/null
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4_call3", "bench.json")).read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["config"]["device_ms"], d["roofline"]["alu"]["frac"])
PY
