#!/bin/bash
# Round-2 GPU call 12: hardware queue count (GPU_MAX_HW_QUEUES) vs the number of concurrent job streams of a proof
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c12
mkdir -p $OUT
export TMPDIR=/tmp
for q in 4 8 16; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python tools/profile_suite.py mimc 40
  GPU_MAX_HW_QUEUES=$q python tools/profile_suite.py proof 20 7 12 | grep create_proof
done > $OUT/hwq.txt 2>&1
cat $OUT/hwq.txt
