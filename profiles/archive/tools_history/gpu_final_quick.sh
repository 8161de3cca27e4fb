#!/bin/bash
# Final run after the last kernel change (sort scatter): the -m gpu suite without the C5-scale file (that file ran green
# on the previous commit, profiles/r2_final_gputests.txt; its MSM cases are re-run here), bench.py, kernel stats, sizes
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2final5
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 600 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_scale.py > $OUT/gputests_noscale.txt 2>&1; tail -2 $OUT/gputests_noscale.txt
timeout 300 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "msm_c5 and (1-23 or 2-20) or density" > $OUT/gputests_scale_subset.txt 2>&1; tail -1 $OUT/gputests_scale_subset.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o p -- $BENCH > $OUT/prof_bench.log 2>&1
python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1.txt 2>&1
python tools/profile_suite.py sizes 1 22 26 > $OUT/sizes_g1_large.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
