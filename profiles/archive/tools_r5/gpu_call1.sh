#!/bin/bash
# round 5, call 1: where do the FFT's wave cycles go (PMC split), and same-box A/B of the experimental kernel builds
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c1
mkdir -p $OUT
export TMPDIR=/tmp
export BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1
LIBS="lib lib_g lib_gl lib_gm lib_glm lib_t512 lib_t512lm lib_notw"
for l in $LIBS; do
  if [ $l != lib_notw ]; then
    BELLMAN_HIP_LIB=bellman_amd/$l/libbellman_hip.so timeout 300 python tools/r5/fft_parity_quick.py > $OUT/parity_$l.txt 2>&1
    echo "$l: $(tail -1 $OUT/parity_$l.txt)"
  fi
done
for rep in 1 2; do
  for l in $LIBS; do
    for sz in "20 10" "22 10" "24 5"; do
      echo "== $l rep $rep" >> $OUT/fft_timing.txt
      BELLMAN_HIP_LIB=bellman_amd/$l/libbellman_hip.so timeout 120 python tools/profile_suite.py fft $sz >> $OUT/fft_timing.txt 2>&1
    done
  done
done
grep -A4 '==' $OUT/fft_timing.txt | grep 'log_n=22\|==' | paste - - - - - | awk '{print $2, $3, $7, $18, $29, $40}' | tail -20
pmc() {  # pmc <tag> <lib> <counters...>
  tag=$1; lib=$2; shift; shift
  BELLMAN_HIP_LIB=bellman_amd/$lib/libbellman_hip.so rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_${lib}_$tag -o p -- python tools/profile_suite.py fft 22 5 > $OUT/pmc_${lib}_$tag.log 2>&1
}
for lib in lib lib_t512lm; do
  pmc a $lib SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  pmc b $lib SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_WAVES
  pmc c $lib SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pmc d $lib TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
  pmc e $lib TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
  pmc f $lib TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  pmc g $lib FETCH_SIZE
  pmc h $lib WRITE_SIZE
  pmc i $lib TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_UTCL1_TRANSLATION_MISS_sum
done
python - <<'PY'
import collections, csv, glob, json, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5c1")
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
for d in $OUT/pmc_*/; do rm -rf $d; done
timeout 900 python -m pytest tests/test_gpu_reference_properties.py tests/test_gpu_fft_extremes.py -x -q --durations=8 > $OUT/new_tests.txt 2>&1; tail -15 $OUT/new_tests.txt
du -sh $OUT
