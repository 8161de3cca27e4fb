#!/bin/bash
# round 4, call 22: deferred reductions inside a proof (BH_MSM_DEFER_REDUCE, BELLMAN_HIP_PROOF_DEFER=1): every multiexp enqueues up to
# its bucket accumulation, merges + reductions of all jobs go behind the last accumulation - so that no latency-bound reduction wave
# takes a register slot from an accumulation.  Parity of a 2^20 proof with the switch on, then the A/B.
# (the switch this run measured was removed afterwards: tools/experiments/r4_deferred_reductions.patch)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c22; mkdir -p $O
( BELLMAN_HIP_PROOF_DEFER=1 timeout 600 python -m pytest tests/test_gpu_groth16.py -q -x -k "chain_2_20 or mimc" ) > $O/parity.txt 2>&1; tail -1 $O/parity.txt
for i in 1 2; do
for d in 0 1; do
  BELLMAN_HIP_PROOF_DEFER=$d python tools/profile_suite.py proof 20 7 12 2>&1 | grep create_proof | sed "s/^/defer=$d /" >> $O/ab.txt
done
done
cat $O/ab.txt
BELLMAN_HIP_PROOF_DEFER=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python tools/profile_suite.py proof 20 3 1 > $O/trace.log 2>&1
f=$(ls $O/trace/*kernel_trace.csv $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); (head -1 $f; tail -3000 $f) > $O/proof_trace.csv; rm -rf $O/trace
python tools/proof_timeline.py $O/proof_trace.csv "deferred reductions" | head -24
