cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/r04/x3_bww_check.py > gpurun_out/r04/x3_bww_check.log 2>&1; echo "check rc=$?"; tail -14 gpurun_out/r04/x3_bww_check.log
timeout -k 10 300 python tools/conv_bench.py bwd_weight --f32x3 > gpurun_out/r04/x3_bww_bench.log 2>&1 && tail -18 gpurun_out/r04/x3_bww_bench.log
M355_F32X3_BWW=0 timeout -k 10 300 python tools/conv_bench.py bwd_weight --f32x3 > gpurun_out/r04/x3_bww_bench_off.log 2>&1 && tail -18 gpurun_out/r04/x3_bww_bench_off.log
