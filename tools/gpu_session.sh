#!/bin/bash
# One gpurun call: GPU tests, smoke, bench, rocprof kernel stats.  Continues past ordinary test
# failures (exit 1) but stops for good if a step was killed or timed out (exit 124/137/139...).
# usage: tools/gpu_session.sh [steps...]   steps: tests tests_all smoke bench benchq arch archprof prof   (default: all)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
STEPS="${*:-tests smoke bench prof}"
export TMPDIR=/tmp
run() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name: $* ===" | tee -a $OUT/session.log
  timeout -k 10 "$to" "$@" > $OUT/$name.log 2>&1
  local rc=$?
  echo "=== $name rc=$rc ===" | tee -a $OUT/session.log
  tail -n 25 $OUT/$name.log
  if [ $rc -ge 124 ]; then echo "step $name was killed (rc=$rc): stopping" | tee -a $OUT/session.log; exit $rc; fi
  return $rc
}
: > $OUT/session.log
rocminfo 2>/dev/null | grep -m1 -E "gfx9" >> $OUT/session.log
for s in $STEPS; do
  case $s in
    tests) run tests 900 python -m pytest tests -m gpu -q -x --timeout=600 ${PYTEST_ARGS:-} ;;
    tests_all) run tests 900 python -m pytest tests -m gpu -q --timeout=600 ${PYTEST_ARGS:-} ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 ${BENCH_ARGS:-} ;;
    benchq) run benchq 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} ;;
    arch) run arch 600 python tools/arch_bench.py ;;
    archprof) run archprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/archprof -- python tools/arch_bench.py
          find $OUT/archprof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/arch_kernel_stats.csv
          head -n 16 $OUT/arch_kernel_stats.csv 2>/dev/null ;;
    pmc) run pmc 900 bash tools/pmc_collect.sh ;;
    prof) run prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer
          find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/kernel_stats.csv
          head -n 40 $OUT/kernel_stats.csv 2>/dev/null ;;
  esac
done
echo "session done" | tee -a $OUT/session.log
