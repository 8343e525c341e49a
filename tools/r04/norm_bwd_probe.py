"""Does the 256 MB memory-side cache serve the second read of a normalisation backward when the two passes run channel block
by channel block (pass 1 / reduce / pass 2 on 67 MB at a time) instead of tensor by tensor (268 MB)?  c8 and fp32, level 0 of cfg2."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps, NormDesc, _p  # noqa: E402
hip = RawOps("hip")
L = hip.lib

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (Cc, sp, G) in [(32, 128, 8), (64, 64, 8), (128, 32, 8)]:
    S = sp ** 3
    CB = Cc // 8
    x16 = torch.randn((1, CB, S, 8), device="cuda").bfloat16()
    dy16 = torch.randn((1, CB, S, 8), device="cuda").bfloat16()
    dx16 = torch.empty_like(x16)
    mean = torch.zeros(G, device="cuda"); rstd = torch.ones(G, device="cuda")
    gamma = torch.ones(Cc, device="cuda"); beta = torch.zeros(Cc, device="cuda")
    dg = torch.empty(Cc, device="cuda"); db = torch.empty(Cc, device="cuda")
    d = NormDesc(1, Cc, S, G, 1, 1e-5, 0.0, 0, 0, 0)
    ws = torch.empty(int(L.m355_norm_workspace(C.byref(d))), dtype=torch.uint8, device="cuda")
    st = hip._stream()
    def full():
        L.m355_norm_act_bwd_c8(C.byref(d), _p(x16), 0, _p(dy16), 0, None, 0, sp, sp, sp, _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx16), 0,
                               _p(dg), _p(db), 1, C.c_float(1.0), 1, _p(ws), ws.numel(), st)
    gpb = G // CB if G >= CB else 1          # groups per c8 block
    d1 = NormDesc(1, 8, S, max(1, G // CB), 1, 1e-5, 0.0, 0, 0, 0)
    def per_block():
        for cb in range(CB):
            o = cb * S * 16
            L.m355_norm_act_bwd_c8(C.byref(d1), C.c_void_p(x16.data_ptr() + o), 0, C.c_void_p(dy16.data_ptr() + o), 0, None, 0, sp, sp, sp,
                                   C.c_void_p(mean.data_ptr() + 4 * cb * gpb), C.c_void_p(rstd.data_ptr() + 4 * cb * gpb),
                                   C.c_void_p(gamma.data_ptr() + 32 * cb), C.c_void_p(beta.data_ptr() + 32 * cb),
                                   C.c_void_p(dx16.data_ptr() + o), 0, C.c_void_p(dg.data_ptr() + 32 * cb), C.c_void_p(db.data_ptr() + 32 * cb),
                                   1, C.c_float(1.0), 1, _p(ws), ws.numel(), st)
    elems = Cc * S
    tf, tb = timeit(full), timeit(per_block)
    print(f"c8   C={Cc:3d} S={sp}^3: whole tensor {tf:7.1f} us ({elems * 10 / tf / 1e6:5.2f} TB/s on 10 B/elem)   block by block {tb:7.1f} us ({elems * 10 / tb / 1e6:5.2f} TB/s, {CB} x 3 launches)")
    # fp32
    x = torch.randn((1, Cc, sp, sp, sp), device="cuda"); dy = torch.randn_like(x); dx = torch.empty_like(x)
    df = NormDesc(1, Cc, S, G, 1, 1e-5, 0.0, 0, 0, 0)
    wsf = torch.empty(int(L.m355_norm_workspace(C.byref(df))), dtype=torch.uint8, device="cuda")
    def full32():
        L.m355_norm_act_bwd(C.byref(df), _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx), _p(dg), _p(db), 1, _p(wsf), wsf.numel(), st)
    cg = Cc // G
    dg1 = NormDesc(1, cg, S, 1, 1e-5 and 1, 1e-5, 0.0, 0, 0, 0)
    def per_group():
        for g in range(G):
            o = g * cg * S * 4
            L.m355_norm_act_bwd(C.byref(dg1), C.c_void_p(x.data_ptr() + o), C.c_void_p(dy.data_ptr() + o), C.c_void_p(mean.data_ptr() + 4 * g),
                                C.c_void_p(rstd.data_ptr() + 4 * g), C.c_void_p(gamma.data_ptr() + 4 * g * cg), C.c_void_p(beta.data_ptr() + 4 * g * cg),
                                C.c_void_p(dx.data_ptr() + o), C.c_void_p(dg.data_ptr() + 4 * g * cg), C.c_void_p(db.data_ptr() + 4 * g * cg), 1, _p(wsf), wsf.numel(), st)
    tf, tb = timeit(full32), timeit(per_group)
    print(f"fp32 C={Cc:3d} S={sp}^3: whole tensor {tf:7.1f} us ({elems * 20 / tf / 1e6:5.2f} TB/s on 20 B/elem)   group by group {tb:7.1f} us ({elems * 20 / tb / 1e6:5.2f} TB/s, {G} x 3 launches)")
