#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r04
timeout -k 10 200 python tools/conv_bench.py fwd bwd_data --f32x3 2>&1 | grep -v amdgpu | tee gpurun_out/r04/x3_fwd_ab.log
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "f32x3 or 3x3x3 or pack" 2>&1 | tail -2
