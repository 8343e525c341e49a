#!/bin/bash
# Round-4 measurement batch (one gpurun call per part): everything lands in gpurun_out/r04m/, the summaries that are
# judged are then copied into profiles/r04_* by tools/r04_collect.py.   usage: tools/r04_measure.sh a|b|c
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
step() { local name=$1 to=$2; shift 2; echo "=== $name" ; timeout -k 10 "$to" "$@" > $O/$name.log 2>&1; local rc=$?; echo "=== $name rc=$rc"; [ $rc -ge 124 ] && exit $rc; return 0; }
stats() { local name=$1; shift; step $name 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o s -- "$@"; find $O/$name -name "s_kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/${name}_kernel_stats.csv; }
case "${1:-a}" in
  a)
    step bench_fp32 600 python bench.py --steps 20 --warmup 5
    step bench_bf16 300 python bench.py --steps 20 --warmup 5 --precision bf16
    step bench_fp16 300 python bench.py --steps 20 --warmup 5 --precision fp16 --no-cpu-baseline
    step bench_cfg4 300 python bench.py --workload cfg4 --steps 3 --warmup 1
    step bench_cfg4_bf16 300 python bench.py --workload cfg4 --steps 5 --warmup 1 --precision bf16 --no-cpu-baseline
    M355_FORCE_DDP=1 step bench_cfg3_rccl1 300 python bench.py --workload cfg3 --bucket-dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
    stats prof_fp32 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg3
    stats prof_fp32_mfma python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg3 --precision fp32_mfma
    stats prof_bf16 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision bf16
    stats prof_bf16_infer python3 tools/infer_profile.py bf16 20
    ;;
  b)
    step pmc 900 bash tools/pmc_collect.sh fp32
    step pmc_fp32_mfma 900 bash tools/pmc_collect.sh fp32_mfma
    step pmc_bf16 900 bash tools/pmc_collect.sh bf16
    step pmc_fp16 900 bash tools/pmc_collect.sh fp16
    step pmc_sq 600 bash tools/pmc_sq.sh
    ;;
  c)
    step arch 300 python tools/arch_bench.py
    step arch_fp32_mfma 300 python tools/arch_bench.py all fp32_mfma
    step arch_bf16 300 python tools/arch_bench.py all bf16
    step layers_cfg2 200 python tools/layer_table.py cfg2
    step conv_fp32 200 python tools/conv_bench.py --f32x3
    step conv_fp32_mfma 200 python tools/conv_bench.py
    step x3_accuracy 300 bash -c "python tools/r04/x3_check.py; python tools/r04/x3_bww_check.py; python tools/r04/split_probe.py"
    step conv_bf16 200 python tools/conv_bench.py --bf16
    step convt_c8 200 python tools/convt_bench_c8.py
    step sliding 300 python tools/sliding_window_bench.py
    step c8_probe 400 python tools/c8_train_probe.py cfg2 cfg5
    step ceiling 200 bash -c "./tools/micro/h16_loop; python tools/r04/gemm_peak.py; ./tools/micro/f32_loop"
    step host 300 python tools/host_overhead_train.py fp32 fp32_mfma bf16 fp16
    ( echo "cfg2 train step only (rocprofv3 kernel stats of bench.py --no-infer, 12 train steps), fp32"; bash tools/train_breakdown.sh fp32 | grep -v "^W20\|^{";
      echo; echo "fp32_mfma"; bash tools/train_breakdown.sh fp32_mfma | grep -v "^W20\|^{";
      echo; echo "bf16"; bash tools/train_breakdown.sh bf16 | grep -v "^W20\|^{";
      echo; echo "fp16"; bash tools/train_breakdown.sh fp16 | grep -v "^W20\|^{" ) > $O/train_breakdown.log 2>&1
    ;;
esac
echo done
