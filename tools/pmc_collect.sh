#!/bin/bash
# HBM traffic of the bench kernels from PMC counters: separate passes per counter, --kernel-trace only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/calib_$ctr -- python tools/pmc_calib.py > $O/calib_$ctr.log 2>&1 || exit 1
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/bench_$ctr -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer > $O/bench_$ctr.log 2>&1 || exit 1
done
find $O -name "*counter_collection.csv" | xargs ls -la
