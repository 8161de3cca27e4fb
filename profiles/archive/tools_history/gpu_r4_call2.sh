#!/bin/bash
# Round 4, GPU call 2: G2 bucket accumulation on lane pairs (csrc/fp2pair.cuh) - group law + multiexp parity, then the
# stage times beside the one-lane-per-point kernel (BH_MSM_G2_SINGLE_LANE = 16) and the lane triples (32).
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_call2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "g2 or G2 or group_law or msm" > $OUT/parity.txt 2>&1
echo "parity: $(tail -1 $OUT/parity.txt)"
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "2-20 or 2-22" > $OUT/scale.txt 2>&1
echo "scale: $(tail -1 $OUT/scale.txt)"
for rep in 1 2; do
  for fl in 0 16 32 256; do
    BH_SUITE_FLAGS=$fl python tools/profile_suite.py sizes 2 16 20 > $OUT/g2_flags${fl}_$rep.txt 2>&1
  done
done
for f in $OUT/g2_flags*.txt; do echo "== $(basename $f)"; cat $f; done
python tools/profile_suite.py sizes 1 18 20 > $OUT/g1.txt 2>&1; cat $OUT/g1.txt
