#!/bin/bash
# Round 3, last call (<= 6 GPU-minutes): the Montgomery reduction with whole columns (ff.cuh, Radix30::NOSPLIT) on the
# device - parity first, then the bench line, the kernel statistics of the timed-steps-only command, FFT / G2 timings.
# Every step is bounded; steps that no longer fit the budget are skipped.
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ns
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
left() { echo $(( ${BUDGET:-330} - ($(date +%s) - T0) )); }
step() {  # step <seconds needed> <name> <cmd...>
  need=$1; name=$2; shift 2
  if [ $(left) -lt $need ]; then echo "SKIP $name ($(left) s left)" | tee -a $OUT/steps.txt; return; fi
  s=$(date +%s)
  timeout $need "$@"
  echo "$name rc=$? $(( $(date +%s) - s )) s" | tee -a $OUT/steps.txt
}
step 100 parity bash -c "python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/parity.txt 2>&1; tail -2 $OUT/parity.txt"
step 150 bench bash -c "python bench.py --no-cpu-baseline --c5-proof-log-n 0 > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json"
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-proof --timed-steps-only"
step 70 prof_bench bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o p -- $BENCH > $OUT/prof_bench.log 2>&1"
step 30 fft bash -c "python tools/profile_suite.py fft 22 10 > $OUT/fft.txt 2>&1; cat $OUT/fft.txt"
step 40 g2 bash -c "python tools/profile_suite.py msm 2 19 5 > $OUT/g2_2p19.txt 2>&1; cat $OUT/g2_2p19.txt"
step 60 groth16 bash -c "python -m pytest tests/test_gpu_groth16.py tests/test_gpu_r1cs.py -m gpu -x -q -k 'not 2_20' > $OUT/groth16.txt 2>&1; tail -2 $OUT/groth16.txt"
step 40 sizes bash -c "python tools/profile_suite.py sizes 1 14 20 > $OUT/sizes_g1.txt 2>&1; tail -3 $OUT/sizes_g1.txt"
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/steps.txt; echo "total $(( $(date +%s) - T0 )) s"
