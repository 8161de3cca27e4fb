#!/bin/bash
# Round-2 final GPU run: the whole -m gpu suite, smoke, bench.py (default flags), and the rocprofv3 evidence that
# DESIGN.md / bench.py cite (kernel stats of the timed-steps-only bench command, PMC passes of the dominant kernel)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2final4
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 1700 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/gputests.txt 2>&1; tail -25 $OUT/gputests.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o p -- $BENCH > $OUT/prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_valu -o p -- $BENCH > $OUT/pmc_valu.log 2>&1
for wl in "fft 22 5" "msm 2 19 5" "msm 1 14 10" "msm 2 16 10" "mimc 10"; do
  tag=$(echo $wl | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o p -- python tools/profile_suite.py $wl > $OUT/prof_$tag.log 2>&1
done
python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1.txt 2>&1
python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2.txt 2>&1
python tools/profile_suite.py sizes 1 22 26 > $OUT/sizes_g1_large.txt 2>&1
python tools/profile_suite.py fft 20 10 > $OUT/fft.txt 2>&1; python tools/profile_suite.py fft 22 10 >> $OUT/fft.txt 2>&1; python tools/profile_suite.py fft 24 5 >> $OUT/fft.txt 2>&1
python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
