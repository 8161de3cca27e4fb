#!/bin/bash
# Round-2 GPU call 1: C5-scale parity tests + baseline per-workload kernel statistics and PMC passes
# of the round-1 kernels that had no rocprof evidence (G2 accumulate, the reductions, the NTT).
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c1
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -s --durations=0 > $OUT/scale_tests.txt 2>&1
tail -5 $OUT/scale_tests.txt
python tools/profile_suite.py sizes 1 10 20 > $OUT/sizes_g1.txt 2>&1
python tools/profile_suite.py sizes 2 10 20 > $OUT/sizes_g2.txt 2>&1
python tools/profile_suite.py mimc 20 > $OUT/mimc.txt 2>&1
python tools/profile_suite.py fft 20 10 > $OUT/fft.txt 2>&1
python tools/profile_suite.py fft 22 10 >> $OUT/fft.txt 2>&1
python tools/profile_suite.py fft 24 5 >> $OUT/fft.txt 2>&1
for wl in "msm 2 19 5" "msm 1 14 10" "msm 2 16 10" "msm 1 20 5" "fft 22 5" "mimc 10"; do
  tag=$(echo $wl | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o p -- python tools/profile_suite.py $wl > $OUT/prof_$tag.log 2>&1
  rm -f $OUT/prof_$tag/*/*.db $OUT/prof_$tag/*.db
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $OUT/pmc_g2_a -o p -- python tools/profile_suite.py msm 2 19 2 > $OUT/pmc_g2_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_g2_b -o p -- python tools/profile_suite.py msm 2 19 2 > $OUT/pmc_g2_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_fft_a -o p -- python tools/profile_suite.py fft 22 2 > $OUT/pmc_fft_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_fft_b -o p -- python tools/profile_suite.py fft 22 2 > $OUT/pmc_fft_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fft_fetch -o p -- python tools/profile_suite.py fft 22 2 > $OUT/pmc_fft_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_fft_write -o p -- python tools/profile_suite.py fft 22 2 > $OUT/pmc_fft_write.log 2>&1
find $OUT -name "*.db" -delete
du -sh $OUT
