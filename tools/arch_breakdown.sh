#!/bin/bash
# Kernel-time breakdown of one of the reference's production architectures (tools/arch_bench.py): rocprofv3 kernel
# stats grouped by family, per train step + per inference forward.   usage: tools/arch_breakdown.sh msseg2|dmri_hippo [fp32|bf16]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
A=${1:-dmri_hippo}; P=${2:-fp32}; O=gpurun_out/arch_$A; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 tools/arch_bench.py $A $P > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
tail -1 $O/run.log
# arch_bench: 13 train steps + 13 no-grad forwards; a forward is ~1/3 of a step's kernels, so "per step" below is
# (train + infer) / 13
python3 tools/train_breakdown.py "$(find $O -name 's_kernel_stats.csv' | head -1)" 13
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/**/s_kernel_stats.csv", recursive=True)[0])))
for r in rows[:24]:
    print(f"  {r['Name'][:86]:86s} {int(r['Calls'])/13:6.1f}/it {float(r['TotalDurationNs'])/13e6:7.3f} ms avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
