cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profbf -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer --precision bf16 > gpurun_out/profbf.log 2>&1
find gpurun_out/profbf -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/bf16_kernel_stats.csv
