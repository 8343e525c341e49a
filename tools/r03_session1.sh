#!/bin/bash
# round-3 first GPU session: whole GPU suite, then the new bench entry points on one GPU
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
O=gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout=900 -x > $O/tests.log 2>&1; rc=$?
tail -n 15 $O/tests.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $O/bench_fp32.json 2> $O/bench_fp32.err; rc=$?
tail -c 600 $O/bench_fp32.json; [ $rc -ne 0 ] && tail -n 20 $O/bench_fp32.err
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python bench.py --workload cfg4 --steps 3 --warmup 1 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; rc=$?
tail -c 1500 $O/bench_cfg4.json; [ $rc -ne 0 ] && tail -n 20 $O/bench_cfg4.err
[ $rc -ge 124 ] && exit $rc
M355_FORCE_DDP=1 timeout -k 10 300 python bench.py --workload cfg3 --bucket-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg3_ddp1.json 2> $O/bench_cfg3_ddp1.err; rc=$?
tail -c 900 $O/bench_cfg3_ddp1.json; [ $rc -ne 0 ] && tail -n 20 $O/bench_cfg3_ddp1.err
exit 0
