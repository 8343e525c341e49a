#!/bin/bash
# Per-step kernel-time breakdown of the TRAIN step only (no inference pass): rocprofv3 kernel stats of
# `bench.py --no-infer`, grouped by kernel family.   usage: tools/train_breakdown.sh [fp32|bf16|fp16]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
P=${1:-fp32}; O=gpurun_out/breakdown_$P; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 bench.py --steps 8 --warmup 2 --no-infer --no-cpu-baseline --no-cfg3 --precision $P > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
tail -1 $O/run.log | cut -c1-200
python3 tools/train_breakdown.py "$(find $O -name 's_kernel_stats.csv' | head -1)" 12
