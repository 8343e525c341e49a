#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/pmc_collect.sh into profiles/<round>_pmc_traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950: FETCH_SIZE counts coalesced
reads at 1/2 (calibration: tools/pmc_calib.py reads/writes known byte counts; the measured factors are
written into the output next to the per-kernel figures).
usage: python tools/pmc_summarize.py [gpurun_out/pmc] [profiles/r01_pmc_traffic.json]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def demangle(name):
    """rocprofv3 prints some template instantiations of this library mangled (_ZN4m355...; the Itanium names of the
    16-bit types, DF16b / DF16_, are unknown to its demangler): decode name + template arguments ourselves"""
    m = re.match(r"_ZN4m355(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        i = 1
        while i < len(rest) and rest[i] != "E":
            if rest.startswith("DF16b", i):
                args.append("__bf16"); i += 5
            elif rest.startswith("DF16_", i):
                args.append("_Float16"); i += 5
            elif rest.startswith("Lb", i):
                args.append("true" if rest[i + 2] == "1" else "false"); i += 4
            elif rest.startswith("Li", i):
                j = rest.index("E", i)
                args.append(rest[i + 2:j].replace("n", "-")); i = j + 1
            else:
                args.append("?"); break
    return f"void m355::{base}<{', '.join(args)}>" if args else f"m355::{base}"


def load(pattern):
    per = defaultdict(lambda: [0.0, 0, 0.0])  # kernel -> [sum counter, launches, sum ms]
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    # gpurun MERGES a call's output into the local gpurun_out/: passes of earlier calls (other PIDs in the file name)
    # pile up next to the new one -- only the newest file of a pass is this measurement
    for fn in files[-1:]:
        for r in csv.DictReader(open(fn)):
            name = re.sub(r"\(.*", "", demangle(r["Kernel_Name"]))
            rec = per[name]
            rec[0] += float(r["Counter_Value"])
            rec[1] += 1
            rec[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return per


def calib(root, ctr):
    """tools/pmc_calib.py launches two kernels with known traffic (far larger than the Infinity Cache):
    copy_channels on 160 Mi floats (float4 path: reads 640 MiB, writes 640 MiB) and add on one element
    fewer (dword path: reads 2 x 640 MiB, writes 640 MiB).  Returns counter*1024 / true bytes per kernel."""
    n4 = 160 * 1024 * 1024
    true = {"copy_channels": (n4 * 4, n4 * 4), "add_kernel": (2 * (n4 - 1) * 4, (n4 - 1) * 4)}
    out = {}
    for name, (csum, n, _) in load(f"{root}/calib_{ctr}/**/*counter_collection.csv").items():
        for key, (rd, wr) in true.items():
            if key in name and n:
                out[key] = csum * 1024 / n / (rd if ctr == "FETCH_SIZE" else wr)
    return out


def source_hash():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_hash as h
    return h()


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
    out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_pmc_traffic.json"
    precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    fetch = load(f"{root}/bench_FETCH_SIZE/**/*counter_collection.csv")
    write = load(f"{root}/bench_WRITE_SIZE/**/*counter_collection.csv")
    kernels = {}
    for name, (fsum, n, ms) in fetch.items():
        if not name.lstrip("void ").startswith("m355::") or name not in write:
            continue
        wsum, wn, _ = write[name]
        fb, wb = 2.0 * fsum * 1024 / n, wsum * 1024 / max(wn, 1)
        kernels[name] = {"launches": n, "hbm_bytes_per_launch": fb + wb, "fetch_bytes_per_launch": fb,
                         "write_bytes_per_launch": wb, "avg_ms_profiled": ms / n}
    kernels = dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
    doc = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on "
                     f"`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --precision {precision}`; bytes = (2*FETCH_SIZE + "
                     "WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads 1/2 for coalesced streams -- calibrated with "
                     "tools/pmc_calib.py)",
           "source_hash": source_hash(),   # bench.py quotes these figures only for exactly these kernel sources
           "calibration": {"FETCH_SIZE": calib(root, "FETCH_SIZE"), "WRITE_SIZE": calib(root, "WRITE_SIZE")},
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in list(kernels.items())[:8]:
        print(f"{k[:70]:70s} {v['launches']:4d} x {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB  {v['avg_ms_profiled']:.3f} ms")


if __name__ == "__main__":
    main()
