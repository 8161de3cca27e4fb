#!/bin/bash
# Round-2 GPU call 4: parity after the merge rewrite (owner finds the run end, G-worker run kernel) and the
# small-proof ordering; plan sweeps with the fixed merge; MiMC timing
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or bases or multiexp or fixed_base or k3" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
timeout 600 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_cpp_api.py tests/test_gpu_generator.py -m gpu -q -x > $OUT/t_groth.txt 2>&1; tail -3 $OUT/t_groth.txt
python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1.txt 2>&1
python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1_notable.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2_notable.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=20 python tools/profile_suite.py sizes 1 17 20 > $OUT/sizes_g1_alltable.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=20 python tools/profile_suite.py sizes 2 16 20 > $OUT/sizes_g2_alltable.txt 2>&1
python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt
# G2: kernel bundle x plan around the switch-over sizes
for ln in 15 16 17 18; do
  BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_FLAGS=32 python tools/tune_msm.py $ln 13,16 0 2 > $OUT/tune_g2_k3_$ln.txt 2>&1
  BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_FLAGS=16 python tools/tune_msm.py $ln 13,16 0 2 > $OUT/tune_g2_single_$ln.txt 2>&1
done
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 python tools/tune_msm.py 20 20 0,32,64 1 > $OUT/tune_table_g1_20.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 BH_FLAGS=16 python tools/tune_msm.py 19 16,20 0,64,128 2 > $OUT/tune_table_g2_19.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 BH_FLAGS=32 python tools/tune_msm.py 19 16 0,128 2 > $OUT/tune_table_g2_19_k3.txt 2>&1
for wl in "msm 1 14 10" "msm 2 16 10" "msm 1 20 5" "mimc 10"; do
  tag=$(echo $wl | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o p -- python tools/profile_suite.py $wl > $OUT/prof_$tag.log 2>&1
  find $OUT/prof_$tag -name "*kernel_trace.csv" -delete
done
BELLMAN_HIP_TABLE_MAX_LOG2=20 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_table_msm_1_20 -o p -- python tools/profile_suite.py msm 1 20 5 > $OUT/prof_table_msm_1_20.log 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=20 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_table_msm_2_19 -o p -- python tools/profile_suite.py msm 2 19 5 > $OUT/prof_table_msm_2_19.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
