#!/bin/bash
# same-box A/B of the conv item order knobs (M355_CONV_CUBE for the fp32 kernels, M355_H16_ORDER for the 16-bit
# ones; bit 0 = (y, z) tiles in 4x4 cubes, bit 1 = channel tile fastest) on an architecture of tools/arch_bench.py
# usage: tools/ab_order.sh [msseg2|dmri_hippo|cfg2] [fp32|bf16|fp16]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=${1:-msseg2}; P=${2:-bf16}
for v in 0 1 2 3 0 3; do
  echo "== $A $P order=$v"; M355_CONV_CUBE=$v M355_H16_ORDER=$v timeout -k 10 200 python tools/arch_bench.py $A $P 2>&1 | tail -1
done
