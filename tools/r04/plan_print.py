#!/usr/bin/env python3
"""prints m355_conv3d_plan for the cfg2 layers: plan_print.py [compute]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps
from tools.conv_bench import CFG2
hip = RawOps("hip")
c = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, ci, co, sp in CFG2:
    print(name, ci, co, sp, "fwd", hip.conv_plan((1, ci, sp, sp, sp), co, compute=c, which=0), "bwd_data",
          hip.conv_plan((1, ci, sp, sp, sp), co, compute=c, which=1), "bww", hip.conv_plan((1, ci, sp, sp, sp), co, compute=c, which=2))
