#!/bin/bash
# Round 3, very last call (what is left of the GPU budget): the rest of the -m gpu suite on the whole-column multiplier
# commit (the parity, groth16 and r1cs files ran in tools/gpu_r3_nosplit.sh), the VALU counter pass of the bench command,
# MiMC and small sizes; then, only if the budget allows, the 2^24-constraint proof test.
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ns2
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
left() { echo $(( ${BUDGET:-265} - ($(date +%s) - T0) )); }
step() {  # step <seconds needed> <name> <cmd...>
  need=$1; name=$2; shift 2
  if [ $(left) -lt $need ]; then echo "SKIP $name ($(left) s left)" | tee -a $OUT/steps.txt; return; fi
  s=$(date +%s)
  timeout $need "$@"
  echo "$name rc=$? $(( $(date +%s) - s )) s" | tee -a $OUT/steps.txt
}
step 150 tests_rest bash -c "python -m pytest tests/test_gpu_round3.py tests/test_gpu_generator.py tests/test_gpu_proof_sharded.py tests/test_cpp_api.py tests/test_gpu_bench_smoke.py -m gpu -x -q > $OUT/tests_rest.txt 2>&1; tail -2 $OUT/tests_rest.txt"
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only"
step 30 pmc_valu bash -c "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_valu -o p -- $BENCH > $OUT/pmc_valu.log 2>&1"
step 20 mimc bash -c "python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt"
step 25 sizes bash -c "python tools/profile_suite.py sizes 1 10 17 > $OUT/sizes_g1_small.txt 2>&1; python tools/profile_suite.py sizes 2 14 18 > $OUT/sizes_g2.txt 2>&1; tail -2 $OUT/sizes_g2.txt"
step 60 params_io bash -c "python -m pytest tests/test_gpu_params_io.py -m gpu -x -q > $OUT/params_io.txt 2>&1; tail -1 $OUT/params_io.txt"
step 100 proof_2p24 bash -c "python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k test_proof_2_24_config_c5 > $OUT/proof_2p24.txt 2>&1; tail -1 $OUT/proof_2p24.txt"
step 40 scale_small bash -c "python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k 'fft_every or 1-23-False or 2-22-True' > $OUT/scale_small.txt 2>&1; tail -1 $OUT/scale_small.txt"
for f in $OUT/pmc_valu/*counter_collection.csv; do [ -f "$f" ] && python - "$f" > $OUT/pmc_valu_accumulate.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "msm_accumulate_kernel" in k or "ntt_pass" in k:
        agg[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()}, "launches", max(len(v) for v in d.values()))
PY
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
cat $OUT/steps.txt; echo "total $(( $(date +%s) - T0 )) s"
