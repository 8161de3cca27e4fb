#!/bin/bash
# Round-2 GPU call 30: FFT parity after the last edit; hardware queue count 16 vs 32 vs 64 (concurrent proofs, MiMC)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c30
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fft or h_poly or domain" 2>&1 | tail -2
for q in 16 32 64; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python tools/profile_suite.py mimc 40
  GPU_MAX_HW_QUEUES=$q python tools/profile_suite.py proof 20 5 12 | grep create_proof
done > $OUT/hwq.txt 2>&1
cat $OUT/hwq.txt
