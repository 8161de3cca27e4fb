#!/bin/bash
# round 5, call 3: validation of everything new since call 2 (10-bit radix passes, call-sites mode 2, table budget, the
# cheaper extreme-vector test) + the bench blocks they feed + sort timing at 2^23 ... 2^26 with 8- and 10-bit passes
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c3
mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 560 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_fft_extremes.py tests/test_gpu_reference_properties.py \
  "tests/test_gpu_parity.py::test_msm_sort_stages" -x -q --durations=12 > $OUT/tests.txt 2>&1; tail -22 $OUT/tests.txt
echo "tests done [$(( $(date +%s) - t0 )) s]"
timeout 200 python -m pytest "tests/test_gpu_scale.py::test_msm_c5_scale_matches_oracle[1-26-False]" -x -q > $OUT/test_2p26.txt 2>&1; tail -3 $OUT/test_2p26.txt
echo "2^26 done [$(( $(date +%s) - t0 )) s]"
timeout 150 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; tail -3 $OUT/bench.err
echo "bench done [$(( $(date +%s) - t0 )) s]"
timeout 100 python tools/profile_suite.py sizes 1 23 26 > $OUT/sizes_10bit.txt 2>&1
BELLMAN_HIP_SORT_10BIT=0 timeout 100 python tools/profile_suite.py sizes 1 23 26 > $OUT/sizes_8bit.txt 2>&1
cat $OUT/sizes_10bit.txt $OUT/sizes_8bit.txt
echo "all done [$(( $(date +%s) - t0 )) s]"
