#!/bin/bash
# NEXT GPU SESSION, first call: A/B of the fused last line of the G1 mixed addition (-DBH_FUSED_Y3).
# Before calling gpurun, build the experimental library HERE (it travels with the snapshot):
#     make -C bellman_amd/csrc -j8 OUT=../lib_exp EXTRA="-DBH_FUSED_Y3=1 -DBH_FUSED_Y3_G2=1 -DBH_FAST_ZERO=1"
# (G1 and the one-lane G2 accumulate kernel; build two libraries to price them separately)
# Then:  gpurun --timeout 600 -- 'bash tools/gpu_r4_fused_y3.sh'
# Parity first (the experiment has only ever run on the host), then timing against the shipped build.
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_fused_y3
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$GRAFT_REPO_ROOT/bellman_amd/lib_exp/libbellman_hip.so
[ -f $EXP ] || { echo "build lib_exp first"; exit 1; }
BELLMAN_HIP_LIB=$EXP timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16.py -m gpu -x -q > $OUT/parity_exp.txt 2>&1; tail -2 $OUT/parity_exp.txt
for rep in 1 2; do
  python tools/profile_suite.py sizes 1 16 20 > $OUT/sizes_base_$rep.txt 2>&1
  BELLMAN_HIP_LIB=$EXP python tools/profile_suite.py sizes 1 16 20 > $OUT/sizes_exp_$rep.txt 2>&1
done
paste -d'\n' $OUT/sizes_base_2.txt $OUT/sizes_exp_2.txt
for rep in 1 2; do
  python tools/profile_suite.py sizes 2 18 20 > $OUT/sizes_g2_base_$rep.txt 2>&1
  BELLMAN_HIP_LIB=$EXP python tools/profile_suite.py sizes 2 18 20 > $OUT/sizes_g2_exp_$rep.txt 2>&1
done
paste -d'\n' $OUT/sizes_g2_base_2.txt $OUT/sizes_g2_exp_2.txt
BENCH="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only"
$BENCH > $OUT/bench_base.json 2>/dev/null; BELLMAN_HIP_LIB=$EXP $BENCH > $OUT/bench_exp.json 2>/dev/null
python - <<'PY'
import json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4_fused_y3")
for t in ("base", "exp"):
    d = json.loads(open(os.path.join(o, "bench_%s.json" % t)).read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d["config"]["device_ms"])
PY
BELLMAN_HIP_LIB=$EXP rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_exp -o p -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only > $OUT/pmc_exp.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
