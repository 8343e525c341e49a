#!/bin/bash
# kernel-only average durations of a python script: tools/r04/kstats.sh <grep pattern> script.py args...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
pat=$1; shift
O=/tmp/ks_$RANDOM; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 "$@" > $O/run.log 2>&1 || tail -5 $O/run.log
python3 tools/r04/topk.py $(find $O -name s_kernel_stats.csv | head -1) 1 60 | grep -E "$pat"
