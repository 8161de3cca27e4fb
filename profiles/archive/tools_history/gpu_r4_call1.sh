#!/bin/bash
# Round 4, GPU call 1: the never-run kernel experiments of round 3 (DESIGN 8.8 d) beside the shipped build.
#   lib_exp_all  = -DBH_FUSED_Y3=1 -DBH_FUSED_Y3_G2=1   (G1 and one-lane G2 accumulate: Y3 as one fused product)
#   lib_exp_y3   = -DBH_FUSED_Y3=1 only, lib_exp_y3g2 = -DBH_FUSED_Y3_G2=1 only (timing, if present)
# Parity on the build with both switches (they touch different kernels), then stage times alternating with the
# shipped build, then a c = 15 probe of the shipped build (17 windows without a sliver, half the buckets).
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call1
mkdir -p $OUT
export TMPDIR=/tmp
ALL=$GRAFT_REPO_ROOT/bellman_amd/lib_exp_all/libbellman_hip.so
[ -f $ALL ] || { echo "build lib_exp_all first"; exit 1; }
BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1 BELLMAN_HIP_LIB=$ALL timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16.py -m gpu -x -q \
   --deselect tests/test_gpu_groth16.py::test_chain_2_20_config_c4 > $OUT/parity_all.txt 2>&1
echo "parity (both switches): $(tail -1 $OUT/parity_all.txt)"
for rep in 1 2; do
  for tag in base all y3 y3g2; do
    if [ $tag = base ]; then unset BELLMAN_HIP_LIB; else
      L=$GRAFT_REPO_ROOT/bellman_amd/lib_exp_$tag/libbellman_hip.so; [ -f $L ] || continue; export BELLMAN_HIP_LIB=$L; fi
    [ $tag = y3g2 ] || python tools/profile_suite.py sizes 1 18 20 > $OUT/g1_${tag}_$rep.txt 2>&1
    [ $tag = y3 ]   || python tools/profile_suite.py sizes 2 19 20 > $OUT/g2_${tag}_$rep.txt 2>&1
  done
done
unset BELLMAN_HIP_LIB
for f in $OUT/g1_*.txt $OUT/g2_*.txt; do echo "== $(basename $f)"; cat $f; done
python tools/profile_suite.py msm 1 20 8 15 0 > $OUT/c15.txt 2>&1
python tools/profile_suite.py msm 1 20 8 15 32 >> $OUT/c15.txt 2>&1
python tools/profile_suite.py msm 1 20 8 16 0 >> $OUT/c15.txt 2>&1
cat $OUT/c15.txt
BENCH="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only"
$BENCH > $OUT/bench_base.json 2>/dev/null
BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1 BELLMAN_HIP_LIB=$ALL $BENCH > $OUT/bench_all.json 2>/dev/null
python - <<'PY'
import glob, json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4_call1")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["ms_per_step"], d["config"]["device_ms"], d["roofline"]["alu"]["frac"])
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
