#!/bin/bash
# the whole GPU suite in one process, log under gpurun_out/r04
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r04/gpu_suite.log 2>&1; rc=$?
echo "suite rc=$rc"; tail -30 gpurun_out/r04/gpu_suite.log
exit $rc
