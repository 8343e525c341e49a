#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for e in M355_NORM_NT=0 M355_NORM_NT=1 M355_NORM_NT=2 M355_NORM_NT=4 M355_NORM_NT=7 M355_NORM_NT=0; do
  O=/tmp/nt_$RANDOM; rm -rf $O; mkdir -p $O
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 bench.py --steps 8 --warmup 2 --no-infer --no-cpu-baseline --no-cfg3 --precision bf16 > $O/run.log 2>&1
  echo "== $e  $(grep '^{' $O/run.log | python3 -c "import sys,json; print('%.3f ms/step' % json.loads(sys.stdin.read())['ms_per_step'])")"
  python3 tools/r04/topk.py $(find $O -name s_kernel_stats.csv | head -1) 12 70 | grep -E "^total|norm_bwd|conv3_bww_c8_kernel|conv3_h16_kernel<4, 32"
done
