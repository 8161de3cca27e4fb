#!/bin/bash
# round 5, call 2: the 512-thread radix-4 FFT kernel with wave-local steps (default build) against the same kernel with every
# barrier, and the 256-thread radix-8 kernel with wave-local steps; PMC of the default; the new parity tests
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c2
mkdir -p $OUT
export TMPDIR=/tmp
export BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1
nproc > $OUT/nproc.txt
LIBS="lib lib_nowl lib_t256wl"
t0=$(date +%s)
for l in $LIBS; do
  BELLMAN_HIP_LIB=bellman_amd/$l/libbellman_hip.so timeout 200 python tools/r5/fft_parity_quick.py 3 10 11 12 13 14 17 20 22 24 > $OUT/parity_$l.txt 2>&1
  echo "$l: $(tail -1 $OUT/parity_$l.txt)  [$(( $(date +%s) - t0 )) s]"
done
for rep in 1 2; do
  for l in $LIBS; do
    for sz in "20 10" "22 10" "24 5"; do
      echo "== $l rep $rep" >> $OUT/fft_timing.txt
      BELLMAN_HIP_LIB=bellman_amd/$l/libbellman_hip.so timeout 60 python tools/profile_suite.py fft $sz >> $OUT/fft_timing.txt 2>&1
    done
  done
done
echo "timing done [$(( $(date +%s) - t0 )) s]"
pmc() {  # pmc <tag> <lib> <counters...>
  tag=$1; lib=$2; shift; shift
  BELLMAN_HIP_LIB=bellman_amd/$lib/libbellman_hip.so timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_${lib}_$tag -o p -- python tools/profile_suite.py fft 22 5 > $OUT/pmc_${lib}_$tag.log 2>&1
}
pmc a lib SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
pmc b lib SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
pmc g lib FETCH_SIZE
pmc h lib WRITE_SIZE
echo "pmc done [$(( $(date +%s) - t0 )) s]"
python - <<'PY'
import collections, csv, glob, json, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5c2")
res = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ntt_pass_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[os.path.basename(d)] = {k: {"mean": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}
json.dump(res, open(os.path.join(out, "pmc_fft.json"), "w"), indent=1)
for k, v in res.items():
    print(k, {n: round(x["mean"]) for n, x in v.items()})
PY
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
timeout 420 python -m pytest tests/test_gpu_reference_properties.py "tests/test_gpu_fft_extremes.py::test_fft_extreme_vectors[11]" "tests/test_gpu_fft_extremes.py::test_fft_extreme_vectors[12]" "tests/test_gpu_fft_extremes.py::test_fft_extreme_vectors[22]" tests/test_gpu_round4.py::test_fft_table_cache_stays_within_its_budget -x -q --durations=8 > $OUT/new_tests.txt 2>&1; tail -15 $OUT/new_tests.txt
echo "new tests done [$(( $(date +%s) - t0 )) s]"
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --durations=5 > $OUT/parity_tests.txt 2>&1; tail -8 $OUT/parity_tests.txt
echo "all done [$(( $(date +%s) - t0 )) s]"
du -sh $OUT
