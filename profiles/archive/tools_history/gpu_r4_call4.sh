#!/bin/bash
# Round 4, GPU call 4: chunk length K with runs folded inside the accumulation workgroup (longer chunks = fewer runs
# that span three chunks, the case that makes a wavefront do two serial additions)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call4
mkdir -p $OUT
export TMPDIR=/tmp
for K in 0 32 40 48 64 96 128; do
  python tools/profile_suite.py msm 1 20 8 16 $K >> $OUT/k_sweep_g1_20.txt 2>&1
done
for K in 0 32 48 64; do
  python tools/profile_suite.py msm 1 19 8 16 $K >> $OUT/k_sweep_g1_19.txt 2>&1
  python tools/profile_suite.py msm 1 18 8 16 $K >> $OUT/k_sweep_g1_18.txt 2>&1
done
cat $OUT/k_sweep_g1_20.txt $OUT/k_sweep_g1_19.txt $OUT/k_sweep_g1_18.txt
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o p -- $BENCH > $OUT/prof_bench.log 2>&1
python tools/kstats.py $OUT/prof_bench 2>/dev/null | head -30
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
