#!/usr/bin/env python3
"""Timeline of the one-shot workgroups of the 16-bit conv kernel on 32->32 @128^3 (diagnostic build, see h16_stamps.py):
entry / exit time (100 MHz) and hardware id of the first 1024 workgroups -> how many run at a time per CU, gaps between
consecutive workgroups on a CU."""
import ctypes as C, os, sys, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps
hip = RawOps("hip"); L = hip.lib
buf = np.zeros((1024, 8), dtype=np.uint64)
ci, co, sp = 32, 32, 128
x16 = hip.act16_pack(torch.randn(1, ci, sp, sp, sp, device="cuda"), 1)
w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
for _ in range(3):
    hip.conv3d_fwd_h16_c8(x16, ci, (sp, sp, sp), w, compute=1)
torch.cuda.synchronize()
assert L.m355_debug_h16_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
t0 = buf[:, 3].astype(np.int64); t1 = buf[:, 4].astype(np.int64); hw = buf[:, 7]
base = t0.min()
print("first 1024 workgroups: entry spread %.1f us, median lifetime %.2f us (entry->exit), median in-loop cycles %d" %
      ((t0.max() - base) / 100, np.median(t1 - t0) / 100, np.median(buf[:, 6])))
hwid = (hw & 0xffffffff).astype(np.int64); xcc = (hw >> 32).astype(np.int64) & 0xf
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 20 + cu
per = collections.defaultdict(list)
for k, a, b in zip(key, t0 - base, t1 - base):
    per[int(k)].append((int(a), int(b)))
print("distinct CUs seen:", len(per), " workgroups per CU (min/median/max):", min(map(len, per.values())), int(np.median(list(map(len, per.values())))), max(map(len, per.values())))
k = sorted(per)[0]
print("one CU's workgroups (entry us, exit us):", [(a / 100, b / 100) for a, b in sorted(per[k])])
# concurrency on that CU over time
ev = sorted([(a, 1) for a, b in per[k]] + [(b, -1) for a, b in per[k]])
c = 0; busy = collections.Counter(); last = ev[0][0]
for t, d in ev:
    busy[c] += t - last; last = t; c += d
print("time with n workgroups resident on it (us):", {n: v / 100 for n, v in sorted(busy.items())})
print("last exit of the first 1024 workgroups: %.1f us after the first entry" % ((t1.max() - base) / 100))
