#!/bin/bash
# Round 4, GPU call 13: FFT kernel instantiation for one-level tables that loads a thread's eight entries at once (the
# chained loads left 32 % of the wave cycles waiting) - parity, timings beside the two-level build; plan-boundary sweeps
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call13
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k fft > $OUT/round4.txt 2>&1; echo "round4: $(tail -1 $OUT/round4.txt)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft or domain or h_poly" > $OUT/parity.txt 2>&1; echo "parity: $(tail -1 $OUT/parity.txt)"
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "fft_every or above" > $OUT/scale_fft.txt 2>&1; echo "scale fft: $(tail -1 $OUT/scale_fft.txt)"
timeout 600 python -m pytest tests/test_gpu_groth16.py -m gpu -x -q -k "mimc or chain_circuit or golden" > $OUT/groth16.txt 2>&1; echo "groth16: $(tail -1 $OUT/groth16.txt)"
for rep in 1 2; do
  for ln in 20 22 24; do python tools/profile_suite.py fft $ln 10 >> $OUT/fft_$rep.txt 2>&1; done
  BELLMAN_HIP_FFT_ONE_LEVEL=0 python tools/profile_suite.py fft 22 10 >> $OUT/fft_two_level.txt 2>&1
done
cat $OUT/fft_1.txt $OUT/fft_2.txt; echo "two-level tables:"; cat $OUT/fft_two_level.txt
python tools/profile_suite.py proof 20 7 12 > $OUT/proof.txt 2>&1; grep create_proof $OUT/proof.txt
python tools/profile_suite.py tsweep 1 13 18 0,13,16 > $OUT/tsweep_g1.txt 2>&1; cat $OUT/tsweep_g1.txt
python tools/profile_suite.py tsweep 2 11 16 8,13,16 > $OUT/tsweep_g2.txt 2>&1; cat $OUT/tsweep_g2.txt
