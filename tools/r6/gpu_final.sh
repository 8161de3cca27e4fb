#!/bin/bash
# Round-6 final GPU run: smoke, the whole -m gpu suite, bench.py (default flags), and the rocprofv3 evidence DESIGN.md /
# bench.py cite: kernel stats of the timed-steps-only bench command, PMC passes of the dominant kernel (FETCH_SIZE,
# WRITE_SIZE, VALU counters - separate passes, no trace domains beside --kernel-trace; every pass under `timeout`), the FFT
# passes; per-size tables (uniform and boolean-heavy); one 2^20 proof's kernel timeline (chain and boolean circuit).
#   usage: gpu_final.sh [nosuite]     then: python tools/summarize_final.py gpurun_out/r6final r6_final
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6final
mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $1"; }
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
if [ "$1" != "nosuite" ]; then
timeout 1100 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/gputests.txt 2>&1; tail -22 $OUT/gputests.txt
fi
stamp suite
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
stamp bench
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only"
P="timeout 90 rocprofv3"
$P --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o p -- $BENCH > $OUT/prof_bench.log 2>&1
$P --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/pmc_fetch.log 2>&1
$P --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $BENCH > $OUT/pmc_write.log 2>&1
$P --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_valu -o p -- $BENCH > $OUT/pmc_valu.log 2>&1
stamp "bench profiles"
$P --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_g2 -o p -- python tools/profile_suite.py msm 2 19 5 > $OUT/pmc_g2.log 2>&1
$P --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fft -o p -- python tools/profile_suite.py fft 22 5 > $OUT/pmc_fft.log 2>&1
$P --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fft_fetch -o p -- python tools/profile_suite.py fft 22 5 > $OUT/pmc_fft_fetch.log 2>&1
$P --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fft_write -o p -- python tools/profile_suite.py fft 22 5 > $OUT/pmc_fft_write.log 2>&1
for wl in "fft 22 5" "msm 2 19 5" "msm 1 16 10" "msm 2 16 10" "mimc 10"; do
  tag=$(echo $wl | tr ' ' '_')
  $P --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o p -- python tools/profile_suite.py $wl > $OUT/prof_$tag.log 2>&1
done
for cfg in "bool50,1,20,1" "bool90,1,20,1" "bool50,2,19,1"; do
  IFS=, read mix g ln tb <<< "$cfg"
  tag=${mix}_g${g}_${ln}
  (cd /tmp && MIX=$mix $P --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o p -- python $GRAFT_REPO_ROOT/tools/r6/boolean_mix.py $g $ln $tb 0 5 > $OUT/prof_$tag.log 2>&1)
done
stamp "kernel profiles"
timeout 60 python tools/profile_suite.py sizes 1 8 20 > $OUT/sizes_g1.txt 2>&1
timeout 60 python tools/profile_suite.py sizes 2 8 20 > $OUT/sizes_g2.txt 2>&1
timeout 100 python tools/profile_suite.py sizes 1 22 26 > $OUT/sizes_g1_large.txt 2>&1
# (table = 1: a registered vector with its automatic window table - the default plan; "1 20 0 0": the classic plan beside it)
(for a in "1 20 1 0" "1 20 1 1" "1 22 1 0" "1 20 0 0" "1 18 1 0" "1 16 1 0" "2 20 1 0" "2 19 1 1"; do timeout 200 python tools/r6/boolean_mix.py $a 9; done) > $OUT/boolean_mix.txt 2>&1
(timeout 40 python tools/profile_suite.py fft 20 10; timeout 40 python tools/profile_suite.py fft 22 10; timeout 40 python tools/profile_suite.py fft 24 5) > $OUT/fft.txt 2>&1
timeout 40 python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1
for i in 1 2; do timeout 60 python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof >> $OUT/proof.txt; done
$P --kernel-trace --output-format csv -d $OUT/trace -o p -- python tools/profile_suite.py proof 20 3 1 > $OUT/trace.log 2>&1
f=$(ls $OUT/trace/*kernel_trace.csv $OUT/trace/*/*kernel_trace.csv 2>/dev/null | head -1); (head -1 $f; tail -3000 $f) > $OUT/proof_trace.csv; rm -rf $OUT/trace
python tools/proof_timeline.py $OUT/proof_trace.csv > $OUT/proof_timeline.txt 2>&1
stamp "tables"
python - <<'PY'
import collections, csv, glob, json, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6final")
def means(d, sub):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"mean": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}
json.dump({"g2_accumulate_lane_pairs_2p19": means("pmc_g2", "msm_accumulate_kernel<bh::Fp2PairOps"),
           "ntt_pass_2p22": means("pmc_fft", "ntt_pass_kernel"),
           "ntt_pass_2p22_fetch": means("pmc_fft_fetch", "ntt_pass_kernel"),
           "ntt_pass_2p22_write": means("pmc_fft_write", "ntt_pass_kernel")}, open(os.path.join(out, "pmc_extra.json"), "w"), indent=1)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
for d in pmc_g2 pmc_fft pmc_fft_fetch pmc_fft_write; do rm -rf $OUT/$d; done
du -sh $OUT; stamp done
