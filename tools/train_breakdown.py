#!/usr/bin/env python3
"""Groups a rocprofv3 kernel-stats CSV of `bench.py --no-infer` by kernel family and prints ms per train step.
usage: train_breakdown.py <s_kernel_stats.csv> <train steps in the run (warm-up + profiled + timed)>"""
import csv
import sys

FAMILIES = [
    ("conv 3x3x3 fwd/bwd-data", ("conv3_mfma_fwd", "conv3_f32x3", "conv3_h16_kernel", "conv3_valu", "conv3_small", "conv3_direct")),
    ("conv transpose", ("convt_",)),
    ("conv weight gradient", ("bww", "slab_reduce", "dbias")),
    ("conv split-K / stats reduce", ("splitk_reduce",)),
    ("weight packs", ("pack_w3", "pack_wt", "pack_batch", "blur_")),
    ("activation packs (fp32 -> c8)", ("pack_act16", "unpack_act16")),
    ("norm statistics", ("norm_partial", "norm_finalize", "norm_from_partials", "act16_partials", "norm_sums")),
    ("norm+act forward", ("norm_act_fwd", "norm_act_pool", "norm_act_c8")),
    ("norm backward", ("norm_bwd",)),
    ("pool / upsample", ("avgpool", "upsample")),
    ("softmax / loss / eval", ("softmax", "loss", "dice", "argmax", "hybrid")),
    ("optimizer + torch elementwise", ("at::native", "multi_tensor", "elementwise_kernel", "vectorized")),
]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2])
    groups, other = {n: [0.0, 0] for n, _ in FAMILIES}, []
    for r in rows:
        ms, calls = float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])
        for name, keys in FAMILIES:
            if any(k in r["Name"] for k in keys):
                groups[name][0] += ms
                groups[name][1] += calls
                break
        else:
            other.append((ms, calls, r["Name"][:70]))
    total = sum(v[0] for v in groups.values()) + sum(o[0] for o in other)
    for name, (ms, calls) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
        print(f"  {name:34s} {ms / steps:7.3f} ms/step  {calls / steps:6.1f} launches/step  {100 * ms / total:5.1f} %")
    for ms, calls, name in sorted(other, reverse=True)[:8]:
        print(f"  other: {name:60s} {ms / steps:7.3f} ms/step  {calls / steps:6.1f}")
    print(f"  total kernel time {total / steps:.3f} ms/step")


main()
