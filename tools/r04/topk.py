#!/usr/bin/env python3
"""top kernels of a rocprofv3 kernel-stats CSV, per train step: topk.py <csv> <steps> [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print(f"total {tot:.3f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
def short(nm):
    nm = re.sub(r"^void ", "", nm); nm = re.sub(r"m355::", "", nm); nm = re.sub(r"\(.*", "", nm)
    m = re.match(r"_ZN4m355\d+(\w+?)I(.*?)EEv", nm)
    return (m.group(1) + "<" + m.group(2) + ">") if m else nm
for r in rows[:n]:
    print(f"{short(r['Name'])[:64]:64s} {int(r['Calls']) / steps:6.1f}/step  {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms/step  avg {float(r['AverageNs']) / 1e3:7.1f} us")
