set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/ab; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for v in 1 3 1 3; do
  echo "== fp32 cube=$v"; M355_CONV_CUBE=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('infer_ms'), d['roofline']['achieved'])"
done
for v in 0 3 0 3; do
  echo "== bf16 order=$v"; M355_H16_ORDER=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('infer_ms'), d['roofline']['achieved'])"
done
for v in 1 3; do echo "== msseg2 cube=$v"; M355_CONV_CUBE=$v timeout -k 10 200 python tools/arch_bench.py msseg2 2>&1 | tail -2; done
