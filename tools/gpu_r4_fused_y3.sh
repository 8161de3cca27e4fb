#!/bin/bash
# NEXT GPU SESSION, first call: A/B of the host-checked, never-run kernel experiments of DESIGN.md 8.8 (d).
# Before calling gpurun, build the experimental libraries HERE (they travel with the snapshot, ~3 MB each):
#     make -C bellman_amd/csrc -j8 OUT=../lib_exp_y3   EXTRA=-DBH_FUSED_Y3=1        # G1: Y3 as one fused product
#     make -C bellman_amd/csrc -j8 OUT=../lib_exp_y3g2 EXTRA=-DBH_FUSED_Y3_G2=1     # one-lane G2 accumulate kernel
#     make -C bellman_amd/csrc -j8 OUT=../lib_exp_all  EXTRA="-DBH_FUSED_Y3=1 -DBH_FUSED_Y3_G2=1"
# Then:  gpurun --timeout 900 -- 'bash tools/gpu_r4_fused_y3.sh'
# For every library found: parity first (the experiments have only ever run on the host), then G1 / G2 stage times
# beside the shipped build's, alternating.
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_fused_y3
mkdir -p $OUT
export TMPDIR=/tmp
LIBS=$(ls -d $GRAFT_REPO_ROOT/bellman_amd/lib_exp*/libbellman_hip.so 2>/dev/null)
[ -n "$LIBS" ] || { echo "build the lib_exp* libraries first (see the header of this script)"; exit 1; }
for EXP in $LIBS; do
  tag=$(basename $(dirname $EXP))
  BELLMAN_HIP_LIB=$EXP timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16.py -m gpu -x -q > $OUT/parity_$tag.txt 2>&1
  echo "$tag parity: $(tail -1 $OUT/parity_$tag.txt)"
done
for rep in 1 2; do
  python tools/profile_suite.py sizes 1 16 20 > $OUT/g1_base_$rep.txt 2>&1
  python tools/profile_suite.py sizes 2 18 20 > $OUT/g2_base_$rep.txt 2>&1
  for EXP in $LIBS; do
    tag=$(basename $(dirname $EXP))
    BELLMAN_HIP_LIB=$EXP python tools/profile_suite.py sizes 1 16 20 > $OUT/g1_${tag}_$rep.txt 2>&1
    BELLMAN_HIP_LIB=$EXP python tools/profile_suite.py sizes 2 18 20 > $OUT/g2_${tag}_$rep.txt 2>&1
  done
done
for f in $OUT/g1_*_2.txt $OUT/g2_*_2.txt; do echo "== $(basename $f)"; cat $f; done
BENCH="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only"
$BENCH > $OUT/bench_base.json 2>/dev/null
for EXP in $LIBS; do
  tag=$(basename $(dirname $EXP))
  BELLMAN_HIP_LIB=$EXP $BENCH > $OUT/bench_$tag.json 2>/dev/null
done
python - <<'PY'
import glob, json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4_fused_y3")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["value"], d["ms_per_step"], d["config"]["device_ms"], d["roofline"]["alu"]["frac"])
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
