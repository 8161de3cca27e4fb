#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python - <<'P' 2>&1 | grep -v Warning
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
torch.cuda.init()
import bench, bellman_amd
from bellman_amd import _lib
lib = _lib.load()
w = bellman_amd.Worker(0)
print("E fresh:", bench.bench_mimc(w, cpu_baseline=False)["ms_median"])
bench.bench_msm_shape(w, lib, 2, 19)
print("F after G2 2^19 shape:", bench.bench_mimc(w, cpu_baseline=False)["ms_median"])
bench.bench_msm_shape(w, lib, 1, 16)
bench.bench_fft(w, lib)
print("G after more:", bench.bench_mimc(w, cpu_baseline=False)["ms_median"])
import ctypes
lib.bh_ctx_trim(w.ctx)
print("H after trim:", bench.bench_mimc(w, cpu_baseline=False)["ms_median"])
P
