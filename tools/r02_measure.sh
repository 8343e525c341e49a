#!/bin/bash
# Round-2 measurement batch (one gpurun call each part): everything lands in gpurun_out/r02/, the summaries that are
# judged are then copied into profiles/ by hand (see tools/README.md).   usage: tools/r02_measure.sh a|b
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
step() { local name=$1 to=$2; shift 2; echo "=== $name" ; timeout -k 10 "$to" "$@" > $O/$name.log 2>&1; local rc=$?; echo "=== $name rc=$rc"; [ $rc -ge 124 ] && exit $rc; return 0; }
stats() { local name=$1; shift; step $name 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o s -- "$@"; find $O/$name -name "s_kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/${name}_kernel_stats.csv; }
case "${1:-a}" in
  a)
    step bench_fp32 500 python bench.py --steps 10 --warmup 3
    step bench_bf16 300 python bench.py --steps 10 --warmup 3 --precision bf16 --no-cpu-baseline
    step bench_fp16 300 python bench.py --steps 10 --warmup 3 --precision fp16 --no-cpu-baseline
    stats prof_fp32 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    stats prof_bf16 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision bf16
    stats prof_bf16_infer python3 tools/infer_profile.py bf16 20
    ;;
  b)
    step pmc 900 bash tools/pmc_collect.sh
    step arch 300 python tools/arch_bench.py
    step arch_bf16 300 python tools/arch_bench.py all bf16
    step layers_cfg2 200 python tools/layer_table.py cfg2
    step layers_msseg2 200 python tools/layer_table.py msseg2
    step conv_fp32 200 python tools/conv_bench.py
    step conv_bf16 200 python tools/conv_bench.py --bf16
    step sliding 300 python tools/sliding_window_bench.py
    step bww_classes 200 python tools/bww_class_probe.py
    step layers_dmri 200 python tools/layer_table.py dmri_hippo
    ( echo "cfg2 train step only (rocprofv3 kernel stats of bench.py --no-infer, 12 train steps), fp32"; bash tools/train_breakdown.sh fp32 | grep -v "^W20\|^{";
      echo; echo "bf16"; bash tools/train_breakdown.sh bf16 | grep -v "^W20\|^{" ) > $O/train_breakdown.log 2>&1
    ( echo "dmri_hippo, fp32: 13 train steps + 13 no-grad forwards, per iteration"; bash tools/arch_breakdown.sh dmri_hippo fp32 | grep -v "^W20";
      echo; echo "msseg2, fp32"; bash tools/arch_breakdown.sh msseg2 fp32 | grep -v "^W20" ) > $O/arch_breakdown.log 2>&1
    ;;
esac
echo done
