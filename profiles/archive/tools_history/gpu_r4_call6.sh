#!/bin/bash
# Round 4, GPU call 6: window groups (first group's reduction beside the last group's accumulation) - parity, then the
# same-box A/B against one group (flags 512 = BH_MSM_ONE_GROUP)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call6
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q > $OUT/parity.txt 2>&1
echo "parity: $(tail -1 $OUT/parity.txt)"
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "msm" > $OUT/scale.txt 2>&1
echo "scale: $(tail -1 $OUT/scale.txt)"
python tools/profile_suite.py sweep 1 20 0 0 0,512 3 > $OUT/g1_20.txt 2>&1
python tools/profile_suite.py sweep 1 19 0 0 0,512 3 > $OUT/g1_19.txt 2>&1
python tools/profile_suite.py sweep 1 21 0 0 0,512 2 > $OUT/g1_21.txt 2>&1
python tools/profile_suite.py sweep 1 22 0 0 0,512 2 > $OUT/g1_22.txt 2>&1
python tools/profile_suite.py sweep 1 24 0 0 0,512 2 > $OUT/g1_24.txt 2>&1
cat $OUT/g1_*.txt
for fl in 0 512; do
BH_BENCH_FLAGS=$fl python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-proof > $OUT/bench_$fl.json 2>/dev/null
done
python - <<'PY'
import json, os
for fl in (0, 512):
    d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4_call6", "bench_%d.json" % fl)).read().strip().splitlines()[-1])
    print("bench flags", fl, d["value"], d["ms_per_step"], d["config"]["device_ms"], d["config"]["value_with_2_jobs_in_flight"], d["roofline"]["alu"]["frac"])
PY
