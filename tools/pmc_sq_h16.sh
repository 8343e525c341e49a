#!/bin/bash
# SQ counters of the 16-bit conv kernel (separate --pmc passes, --kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/pmcsq16; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python tools/pmc_sq_h16.py > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("gpurun_out/pmcsq16/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "conv3_h16" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"][:60], r.get("Grid_Size"), r.get("LDS_Block_Size"))
        acc[(r["Kernel_Name"][:48], r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[(r["Kernel_Name"][:48], r["Dispatch_Id"])]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc, key=lambda k: int(k[1])):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()})
PY
