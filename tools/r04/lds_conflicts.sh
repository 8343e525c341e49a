#!/bin/bash
# LDS bank conflicts per kernel of a train step: tools/r04/lds_conflicts.sh [fp32|bf16|fp16|fp32_mfma]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
P=${1:-bf16}; O=gpurun_out/ldsc_$P; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-cfg3 --precision $P > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, collections, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.Counter()
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"{'kernel':70s} {'launches':>8s} {'us total':>10s} {'LDS active':>11s} {'conflict':>9s} {'MFMA busy':>9s}")
for k in sorted(acc, key=lambda k: -dur[k])[:28]:
    a = acc[k]; cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
    print(f"{k:70s} {n[k]:8d} {dur[k]:10.0f} {a.get('SQ_LDS_IDX_ACTIVE', 0) / cyc / 256:11.3f} "
          f"{(a.get('SQ_LDS_BANK_CONFLICT', 0) / a['SQ_LDS_IDX_ACTIVE']) if a.get('SQ_LDS_IDX_ACTIVE') else 0:9.3f} {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / cyc:9.3f}")
PY
