#!/bin/bash
# Round-2 GPU call 6: parity with split accumulate / reduce bundles and G2 tables at every size; G1 table sweeps
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c6
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or bases or multiexp or fixed_base or k3" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
timeout 600 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_cpp_api.py -m gpu -q -x > $OUT/t_groth.txt 2>&1; tail -3 $OUT/t_groth.txt
python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2.txt 2>&1
python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1.txt 2>&1
python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt
for ln in 17 18 19 20; do
  BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 python tools/tune_msm.py $ln 16,19,20,22 0 1 > $OUT/tune_table_g1_$ln.txt 2>&1
done
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 python tools/tune_msm.py 22 19,20,22 0 1 > $OUT/tune_table_g1_22.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 python tools/tune_msm.py 19 16 0,64,128 2 > $OUT/tune_table_g2_19.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=22 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_table_msm_2_19 -o p -- python tools/profile_suite.py msm 2 19 5 > $OUT/prof_table_msm_2_19.log 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=22 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_table_msm_1_20 -o p -- python tools/profile_suite.py msm 1 20 5 > $OUT/prof_table_msm_1_20.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
