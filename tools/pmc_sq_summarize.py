#!/usr/bin/env python3
"""Turn the four rocprofv3 --pmc passes of tools/pmc_sq.sh (gpurun_out/pmcsq/p1..p4) into profiles/<round>_pmc_sq_counters.json:
per-launch means of each SQ counter for the two dominant fp32 kernels, the effective clock (GRBM_GUI_ACTIVE / time) and
the MFMA-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / GRBM_GUI_ACTIVE).
usage: python tools/pmc_sq_summarize.py [gpurun_out/pmcsq] [profiles/r03_pmc_sq_counters.json]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summarize import demangle  # noqa: E402


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmcsq"
    out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_pmc_sq_counters.json"
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum, launches]
    dur = defaultdict(lambda: [0.0, 0])
    for p in sorted(glob.glob(f"{root}/p*/")):
        files = sorted(glob.glob(f"{p}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
        for fn in files[-1:]:
            seen = set()
            for r in csv.DictReader(open(fn)):
                name = re.sub(r"\(.*", "", demangle(r["Kernel_Name"]))
                if not any(k in name for k in ("conv3_mfma", "conv3_f32x3", "conv3_bww_x3")):
                    continue
                a = acc[name][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                key = (name, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[name][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    dur[name][1] += 1
    kernels = {}
    for name, ctrs in acc.items():
        rec = {c: v[0] / v[1] for c, v in ctrs.items()}
        rec["dur_us_profiled"] = dur[name][0] / max(dur[name][1], 1)
        if "GRBM_GUI_ACTIVE" in rec:
            # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs
            rec["derived_clock_GHz"] = rec["GRBM_GUI_ACTIVE"] / 8.0 / rec["dur_us_profiled"] / 1e3
            if "SQ_VALU_MFMA_BUSY_CYCLES" in rec:
                rec["derived_mfma_pipe_utilisation"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (rec["GRBM_GUI_ACTIVE"] / 8.0)
        kernels[name] = rec
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_hash
    doc = {"method": "rocprofv3 --pmc (four passes of three counters, --kernel-trace only; tools/pmc_sq.sh) on 3 forward + 3 "
                     "weight-gradient launches of the largest layer (96->32 @128^3) on the split kernels and on the fp32 MFMA kernels; per-launch means; clock = GRBM_GUI_ACTIVE / "
                     "(8 XCDs x time); MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)",
           "source_hash": source_hash(), "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(k[:70], {c: round(x, 4) for c, x in v.items() if c.startswith("derived") or c == "dur_us_profiled"})


if __name__ == "__main__":
    main()
