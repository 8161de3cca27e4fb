#!/bin/bash
# Round-2 GPU call 8: lane cost model slot factors, stall counters of the one-lane-per-point G2 accumulation,
# priority stream for the reduction phase on the 2^20 proof, c=20 at 2^24 with the new merge
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c8
mkdir -p $OUT
export TMPDIR=/tmp
for f in "1,1,2" "2,1,2"; do echo "SUM_SLOTS=$f"; BELLMAN_HIP_SUM_SLOTS=$f python tools/profile_suite.py sizes 1 10 20; done > $OUT/slots_g1.txt 2>&1
for f in "1,1,2" "1,1,1" "1,2,2"; do echo "SUM_SLOTS=$f"; BELLMAN_HIP_SUM_SLOTS=$f python tools/profile_suite.py sizes 2 10 20; done > $OUT/slots_g2.txt 2>&1
cat $OUT/slots_g1.txt $OUT/slots_g2.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_g2_a -o p -- python tools/profile_suite.py msm 2 19 2 > $OUT/pmc_g2_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_g2_b -o p -- python tools/profile_suite.py msm 2 19 2 > $OUT/pmc_g2_b.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/pmc_g2_c -o p -- python tools/profile_suite.py msm 2 19 2 > $OUT/pmc_g2_c.log 2>&1
python tools/profile_suite.py proof 20 5 12 > $OUT/proof_prio.txt 2>&1
BELLMAN_HIP_REDUCE_PRIORITY=1 python tools/profile_suite.py proof 20 5 12 >> $OUT/proof_prio.txt 2>&1
cat $OUT/proof_prio.txt
python tools/tune_msm.py 24 16,20 0,32,64 1 > $OUT/tune_g1_24.txt 2>&1; cat $OUT/tune_g1_24.txt
python tools/tune_msm.py 23 16,20 0,32 1 > $OUT/tune_g1_23.txt 2>&1; cat $OUT/tune_g1_23.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
