#!/bin/bash
# Round-2 GPU call 2: parity of the new kernels (K3 G2, tiny host path, window tables by default for small
# vectors, the two-pass register-radix FFT) + per-workload kernel statistics (csv this time)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c2
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "k3 or field_mul or point_add" > $OUT/t_k3.txt 2>&1; tail -3 $OUT/t_k3.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fft or domain or h_poly" > $OUT/t_fft.txt 2>&1; tail -3 $OUT/t_fft.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "msm or bases or multiexp or fixed_base" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
timeout 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_generator.py tests/test_gpu_params_io.py tests/test_gpu_r1cs.py tests/test_gpu_proof_sharded.py tests/test_cpp_api.py -m gpu -q > $OUT/t_groth.txt 2>&1; tail -3 $OUT/t_groth.txt
for tm in 12 0 16; do
  BELLMAN_HIP_TABLE_MAX_LOG2=$tm python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1_table$tm.txt 2>&1
  BELLMAN_HIP_TABLE_MAX_LOG2=$tm python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2_table$tm.txt 2>&1
done
python tools/profile_suite.py mimc 20 > $OUT/mimc.txt 2>&1
python tools/profile_suite.py fft 20 10 > $OUT/fft.txt 2>&1
python tools/profile_suite.py fft 22 10 >> $OUT/fft.txt 2>&1
python tools/profile_suite.py fft 24 5 >> $OUT/fft.txt 2>&1
cat $OUT/mimc.txt $OUT/fft.txt
for wl in "msm 2 19 5" "msm 1 14 10" "msm 2 16 10" "msm 1 20 5" "msm 1 24 3 20" "msm 1 24 3" "fft 22 5" "mimc 10"; do
  tag=$(echo $wl | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o p -- python tools/profile_suite.py $wl > $OUT/prof_$tag.log 2>&1
  find $OUT/prof_$tag -name "*kernel_trace.csv" -delete
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_g2_a -o p -- python tools/profile_suite.py msm 2 19 2 > $OUT/pmc_g2_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_g2_b -o p -- python tools/profile_suite.py msm 2 19 2 > $OUT/pmc_g2_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fft_a -o p -- python tools/profile_suite.py fft 22 2 > $OUT/pmc_fft_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fft_b -o p -- python tools/profile_suite.py fft 22 2 > $OUT/pmc_fft_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fft_fetch -o p -- python tools/profile_suite.py fft 22 2 > $OUT/pmc_fft_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fft_write -o p -- python tools/profile_suite.py fft 22 2 > $OUT/pmc_fft_write.log 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
# window-table plan sweeps at small sizes (tables rebuilt per c): c x K
for ln in 10 12 14 16; do
  BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 python tools/tune_msm.py $ln 10,12,13,14,16 4,8,16 1 > $OUT/tune_table_g1_$ln.txt 2>&1
done
for ln in 10 14 16; do
  BELLMAN_HIP_TABLE_MAX_LOG2=0 BH_TABLE=1 python tools/tune_msm.py $ln 10,13,16 4,8,16 2 > $OUT/tune_table_g2_$ln.txt 2>&1
done
du -sh $OUT
