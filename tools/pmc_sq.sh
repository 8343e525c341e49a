#!/bin/bash
# SQ counters of the two dominant kernels (separate --pmc passes, --kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/pmcsq; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python tools/pmc_sq_layer.py > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; exit 1; }
done
find $O -name "*counter_collection.csv" | xargs ls -la
