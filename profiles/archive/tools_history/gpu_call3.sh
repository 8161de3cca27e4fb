#!/bin/bash
# Round-2 GPU call 3: parity after the merge-path / plan changes, then plan sweeps (classic vs window table at
# every size, G2 kernel bundles) and MiMC timing
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or bases or multiexp or fixed_base or k3" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
timeout 600 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_proof_sharded.py tests/test_cpp_api.py -m gpu -q -x > $OUT/t_groth.txt 2>&1; tail -3 $OUT/t_groth.txt
python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1.txt 2>&1
python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1_notable.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2_notable.txt 2>&1
python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt
# classic plans for G2: c x K x kernel bundle
for ln in 14 16 17 18 19; do
  BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_FLAGS=16 python tools/tune_msm.py $ln 13,16 16,32,64 2 > $OUT/tune_g2_single_$ln.txt 2>&1
done
for ln in 14 16 17; do
  BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_FLAGS=32 python tools/tune_msm.py $ln 13,16 8,16,32 2 > $OUT/tune_g2_k3_$ln.txt 2>&1
done
# window tables at large sizes
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 python tools/tune_msm.py 20 16,20,22 32,64 1 > $OUT/tune_table_g1_20.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 python tools/tune_msm.py 18 16,20 16,32 1 > $OUT/tune_table_g1_18.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 BH_FLAGS=16 python tools/tune_msm.py 19 16,20 32,64,128 2 > $OUT/tune_table_g2_19.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 BH_FLAGS=16 python tools/tune_msm.py 17 16,20 16,32,64 2 > $OUT/tune_table_g2_17.txt 2>&1
BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 python tools/tune_msm.py 22 20,22 32,64 1 > $OUT/tune_table_g1_22.txt 2>&1
for wl in "msm 1 14 10" "msm 2 16 10" "msm 1 20 5" "mimc 10"; do
  tag=$(echo $wl | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o p -- python tools/profile_suite.py $wl > $OUT/prof_$tag.log 2>&1
  find $OUT/prof_$tag -name "*kernel_trace.csv" -delete
done
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
