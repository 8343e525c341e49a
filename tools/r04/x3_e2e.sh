#!/bin/bash
# fp32x3 end to end: the fp32 parity tests with every fp32 conv forced onto the split kernels, then the bench legs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r04
M355_F32X3=2 timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "not 16bit and not bf16 and not fp16 and not h16 and not c8" > gpurun_out/r04/x3_forced_tests.log 2>&1; echo "forced tests rc=$?"; tail -5 gpurun_out/r04/x3_forced_tests.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04/x3_bench.json 2> gpurun_out/r04/x3_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r04/x3_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/x3_bench.json"))
print("fp32", d["ms_per_step"], d["infer"]["ms_per_step"], d["roofline"]["frac"])
for k in ("cfg3","f32x3"):
    e=d[k]; print(k, e["ms_per_step"], e["infer_ms"], e["roofline"], e.get("max_abs_diff_of_step0_probabilities_vs_fp32"), e["final_loss"], d["final_loss"])
PY
