#!/bin/bash
# Round 6, call 55: 13-bit tables for tiny G2 vectors: parity (small sizes, proofs), MiMC-322 proof, G2 sizes 2^8 ... 2^13
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r6c55_tiny_g2_13bit.txt
: > $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16.py tests/test_gpu_round3.py tests/test_gpu_generator.py tests/test_gpu_proof_sharded.py -q -x -m gpu 2>&1 | tail -3 >> $out
for i in 1 2 3; do timeout 100 python tools/profile_suite.py mimc 30 >> $out 2>&1; done
timeout 100 python tools/profile_suite.py sizes 2 8 13 >> $out 2>&1
