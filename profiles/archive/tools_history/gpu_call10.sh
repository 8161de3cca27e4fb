#!/bin/bash
# Round-2 GPU call 10: FFT with batched loads / stores (parity at every size + timing + kernel stats)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c10
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fft or domain or h_poly" > $OUT/t_fft.txt 2>&1; tail -3 $OUT/t_fft.txt
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -x -k "fft" > $OUT/t_fft_scale.txt 2>&1; tail -3 $OUT/t_fft_scale.txt
for l in 10 14 16 18 20 22 24; do python tools/profile_suite.py fft $l 10; done > $OUT/fft.txt 2>&1; cat $OUT/fft.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fft_22 -o p -- python tools/profile_suite.py fft 22 5 > $OUT/prof_fft_22.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fft -o p -- python tools/profile_suite.py fft 22 2 > $OUT/pmc_fft.log 2>&1
python tools/profile_suite.py mimc 30 > $OUT/mimc.txt 2>&1; cat $OUT/mimc.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
