#!/bin/bash
# round-4 session 1: full GPU suite on the ABI-v3 library + the bench line with the extra cfg3 key
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04s1; mkdir -p $O
step() { local name=$1 to=$2; shift 2; echo "=== $name"; timeout -k 10 "$to" "$@" > $O/$name.log 2>&1; local rc=$?; echo "=== $name rc=$rc"; tail -n 6 $O/$name.log; [ $rc -ge 124 ] && exit $rc; return 0; }
step tests 1000 python -m pytest tests -m gpu -q -x --timeout=600
step bench 500 python bench.py --steps 20 --warmup 5
echo done
