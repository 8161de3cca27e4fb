#!/bin/bash
# Round 4, GPU call 11: G1 base records at a 128-byte stride for the gathers of the bucket accumulation
# (BELLMAN_HIP_BASE_PAD=1) beside the dense 96-byte layout: stage times, FETCH_SIZE of the accumulate kernel
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call11
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3; do
  for pad in 0 1; do
    BELLMAN_HIP_BASE_PAD=$pad python tools/profile_suite.py sweep 1 20 0 0 0 2 >> $OUT/t20_pad$pad.txt 2>&1
  done
done
for pad in 0 1; do
  BELLMAN_HIP_BASE_PAD=$pad python tools/profile_suite.py sweep 1 22 0 0 0 2 >> $OUT/t22_pad$pad.txt 2>&1
  BELLMAN_HIP_BASE_PAD=$pad python tools/profile_suite.py sweep 1 18 0 0 0 2 >> $OUT/t18_pad$pad.txt 2>&1
done
for f in t18_pad0 t18_pad1 t20_pad0 t20_pad1 t22_pad0 t22_pad1; do echo "== $f"; cat $OUT/$f.txt; done
for pad in 0 1; do
  BELLMAN_HIP_BASE_PAD=$pad rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_pad$pad -o p -- python tools/profile_suite.py msm 1 20 5 > $OUT/pmc_pad$pad.log 2>&1
  python - <<PY
import csv, glob
vals = []
for f in glob.glob("$OUT/pmc_fetch_pad$pad/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msm_accumulate_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
print("pad=$pad FETCH_SIZE per accumulate launch (KB): mean %.0f over %d launches" % (sum(vals) / max(1, len(vals)), len(vals)))
PY
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
BELLMAN_HIP_BASE_PAD=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm_2_20 or linearity or fuzz" > $OUT/parity_pad.txt 2>&1; echo "parity (padded): $(tail -1 $OUT/parity_pad.txt)"
