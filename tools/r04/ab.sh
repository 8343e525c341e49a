#!/bin/bash
# same-box A/B of environment knobs on the bench line: tools/r04/ab.sh "<bench args>" "ENV1=.. ENV2=.." "ENV=.." ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
args=$1; shift
for e in "$@"; do
  out=$(env $e python bench.py $args --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
  echo "$e :: $(echo "$out" | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('train %.3f ms  infer %s ms  loss %.6f' % (d['ms_per_step'], ('%.3f' % d['infer']['ms_per_step']) if d.get('infer') else '-', d['final_loss']))")"
done
