#!/bin/bash
# round-3 second session: whole GPU suite with the c8-only training flow, probes, 16-bit benches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
O=gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout=900 > $O/tests2.log 2>&1; rc=$?
tail -n 30 $O/tests2.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 400 python tools/c8_train_probe.py cfg2 cfg5 > $O/probe2.log 2>&1; rc=$?
grep -v Warning $O/probe2.log | tail -n 40
[ $rc -ge 124 ] && exit $rc
for prec in bf16 fp16; do
  timeout -k 10 300 python bench.py --precision $prec --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$prec.json 2> $O/bench_$prec.err; rc=$?
  python - <<P
import json
d=json.loads(open("$O/bench_$prec.json").read().strip().splitlines()[-1])
print("$prec", "train ms/step", round(d["ms_per_step"],3), "infer ms", round(d["infer"]["ms_per_step"],3), d["roofline"]["kernel"], round(d["roofline"]["frac"],3), {k: round(v["ms_per_step"],3) for k,v in d["conv_kernels"].items()}, "loss", d["final_loss"])
P
  [ $rc -ge 124 ] && exit $rc
done
exit 0
