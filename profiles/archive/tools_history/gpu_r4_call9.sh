#!/bin/bash
# Round 4, GPU call 9: FFT one-level tables in TILE order (coalesced whatever the tile shape) - parity, timings
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q > $OUT/round4.txt 2>&1; echo "round4: $(tail -1 $OUT/round4.txt)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft or domain or h_poly" > $OUT/parity.txt 2>&1; echo "parity: $(tail -1 $OUT/parity.txt)"
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "fft" > $OUT/scale_fft.txt 2>&1; echo "scale fft: $(tail -1 $OUT/scale_fft.txt)"
timeout 600 python -m pytest tests/test_gpu_groth16.py -m gpu -x -q -k "mimc or chain_circuit or golden" > $OUT/groth16.txt 2>&1; echo "groth16: $(tail -1 $OUT/groth16.txt)"
for rep in 1 2; do for ln in 20 21 22 24; do python tools/profile_suite.py fft $ln 10 >> $OUT/fft_$rep.txt 2>&1; done; done
BELLMAN_HIP_FFT_ONE_LEVEL=0 python tools/profile_suite.py fft 22 10 > $OUT/fft_two_level.txt 2>&1
cat $OUT/fft_1.txt $OUT/fft_2.txt; echo "two-level tables:"; cat $OUT/fft_two_level.txt
python tools/profile_suite.py sizes 2 13 17 > $OUT/g2.txt 2>&1; cat $OUT/g2.txt
python tools/profile_suite.py proof 20 7 12 > $OUT/proof.txt 2>&1; grep create_proof $OUT/proof.txt
