#!/bin/bash
# the driver's command (default flags), output under gpurun_out/r04
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r04
timeout -k 10 900 python bench.py "$@" > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r04/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/bench_default.json"))
print("headline", d["ms_per_step"], d["infer"]["ms_per_step"], d["dtype"][:40], d["roofline"])
print("conv", d["conv_kernels"])
print("dice", d["dice"])
print("cpu", {k: v for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
for k in ("cfg3","fp32_mfma"):
    e=d.get(k)
    if e: print(k, e["ms_per_step"], e["infer_ms"], e["roofline"]["frac"], e["roofline"]["kernel"], e.get("max_abs_diff_of_step0_probabilities_vs_headline"), e["final_loss"], d["final_loss"])
PY
