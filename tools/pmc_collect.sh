#!/bin/bash
# HBM traffic of the bench kernels from PMC counters: separate passes per counter, --kernel-trace only.
# usage: tools/pmc_collect.sh [fp32|bf16|fp16]   -> gpurun_out/pmc[_<precision>]/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
P=${1:-fp32}
O=gpurun_out/pmc; [ "$P" != fp32 ] && O=gpurun_out/pmc_$P
rm -rf $O; mkdir -p $O
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/calib_$ctr -- python3 tools/pmc_calib.py > $O/calib_$ctr.log 2>&1 || exit 1
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/bench_$ctr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-cfg3 --precision $P > $O/bench_$ctr.log 2>&1 || exit 1
done
find $O -name "*counter_collection.csv" | xargs ls -la
