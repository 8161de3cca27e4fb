#!/bin/bash
# round 6, call 13: UPPER BOUND of a Karatsuba / lazy-reduction Fp2 product on lane pairs: the second product
# of the pair's fe_mul2 cut to half its rows - 1.5 products + one reduction per lane with every exchange and addition free
# (timing-only: results are wrong).  Same box, alternating.
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c13
mkdir -p $OUT
for rep in 1 2 3; do
  for ln in 19 20; do
    echo "shipped: $(timeout 120 python tools/profile_suite.py msm 2 $ln 8 | tail -1)"
    echo "g2half : $(BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1 BELLMAN_HIP_LIB=$GRAFT_REPO_ROOT/bellman_amd/lib_g2half/libbellman_hip.so timeout 120 python tools/profile_suite.py msm 2 $ln 8 | tail -1)"
  done
done 2>&1 | tee $OUT/g2_half_product_upper_bound.txt
