#!/bin/bash
# Round-2 GPU call 9: software-pipelined loads in the one-lane-per-point G2 accumulation (parity + per-size tables),
# G1 slots fix, c=20 chunk sweep at 2^24..2^26
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or bases or multiexp or fixed_base or k3" > $OUT/t_msm.txt 2>&1; tail -3 $OUT/t_msm.txt
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -q -x -k "msm_c5 and (2-20 or 2-22 or 1-23)" > $OUT/t_scale.txt 2>&1; tail -3 $OUT/t_scale.txt
python tools/profile_suite.py sizes 2 14 22 > $OUT/sizes_g2.txt 2>&1; cat $OUT/sizes_g2.txt
python tools/profile_suite.py sizes 1 16 20 > $OUT/sizes_g1.txt 2>&1; cat $OUT/sizes_g1.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_g2 -o p -- python tools/profile_suite.py msm 2 19 2 > $OUT/pmc_g2.log 2>&1
python tools/tune_msm.py 24 20 0,128 1 > $OUT/tune_g1_c20.txt 2>&1
python tools/tune_msm.py 25 20 64,0,256 1 >> $OUT/tune_g1_c20.txt 2>&1
python tools/tune_msm.py 26 20 128,0,512 1 >> $OUT/tune_g1_c20.txt 2>&1
cat $OUT/tune_g1_c20.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
