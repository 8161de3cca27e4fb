#!/bin/bash
# round 4, call 23 (the last minutes of the budget): the GPU tests that exercise what changed after the third final run -
# the error-resolution pass over a 128-byte-stride table (fd45696) and Parameters::write into the caller's buffer (f9b7e4d)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c23; mkdir -p $O
timeout 200 python -m pytest -q -x -m gpu tests/test_gpu_round4.py::test_g1_window_table_at_128_byte_stride tests/test_gpu_params_io.py tests/test_gpu_generator.py "tests/test_gpu_groth16.py" -k "not 2_20" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
