"""Fused W-MSA block kernel (csrc/wmsa_block.hip) at the bench shape (B=8 @1024^2 -> 524,288 tokens, C=192), with an
L2 / Infinity-Cache flush between calls: inference form, training form (saved tensors), and the four launches it replaces."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "small-object-detection-transformers_amd"
ops = importlib.import_module(PKG + ".ops")
L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0"); dt = torch.bfloat16
big = torch.empty(1 << 28, device=dev, dtype=torch.float32)
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
tz = timeit(lambda: big.zero_())
C, HEADS, WS = 192, 12, 8
B, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 256
M = B * H * H
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
qw, qb, pw, pb = r(3 * C, C, sc=0.1), r(3 * C, sc=0.1), r(C, C, sc=0.1), r(C, sc=0.1)
tab, n1w, n1b, n2w, n2b = r(225, HEADS, sc=0.3), 1 + r(C, sc=0.1), r(C, sc=0.1), 1 + r(C, sc=0.1), r(C, sc=0.1)
x = r(M, C).to(dt)
wpk = torch.zeros(ops.wmsa_pack_bytes(C, HEADS, WS, L.BF16) // 2, device=dev, dtype=dt)
ops.wmsa_pack(qw, qb, pw, pb, tab, n1w, n1b, n2w, n2b, wpk, C, HEADS, WS)
xm, xn2, xn1, ao = (torch.empty(M, C, device=dev, dtype=dt) for _ in range(4))
st1, st2 = torch.empty(M, 2, device=dev), torch.empty(M, 2, device=dev)
lsew = torch.empty(M // 64, HEADS, 64, device=dev)
flops = 8.0 * M * C * C + 4.0 * M * 64 * C
for shift in (0, 2):
    def inf(): big.zero_(); ops.wmsa_block_fwd(x, wpk, xm, xn2, None, None, None, None, None, None, B, H, H, C, HEADS, WS, shift)
    def trn(): big.zero_(); ops.wmsa_block_fwd(x, wpk, xm, xn2, st1, st2, xn1, None, lsew, ao, B, H, H, C, HEADS, WS, shift)
    a, b = timeit(inf) - tz, timeit(trn) - tz
    print(f"fused shift={shift}: inference {a:.3f} ms = {flops/a/1e9:.0f} TF/s ({flops/a/1e9/2500:.3f} of 2.5 PF), "
          f"training {b:.3f} ms = {flops/b/1e9:.0f} TF/s ({flops/b/1e9/2500:.3f}); alg. bytes inf {M*C*2*3/a/1e6:.0f} GB/s, "
          f"train {M*C*2*5/b/1e6:.0f} GB/s", flush=True)
# the launches it replaces
qkv = torch.empty(M, 3 * C, device=dev, dtype=dt); lse = torch.empty(M, HEADS, device=dev)
wq, wp_, bt = qw.to(dt), pw.to(dt), tab.t().contiguous()
def unf():
    big.zero_()
    ops.layernorm_fwd(x, n1w, n1b, xn1, st1, M, C)
    ops.gemm_nt([ops.SegSpec(xn1)], wq, qkv, M, 3 * C, C, bias=qb)
    ops.window_attn_fwd(qkv, bt, ao, lse, B, H, H, C, HEADS, WS, 0)
    ops.gemm_nt([ops.SegSpec(ao)], wp_, xm, M, C, C, bias=pb, resid=x)
    ops.layernorm_fwd(xm, n2w, n2b, xn2, st2, M, C)
u = timeit(unf) - tz
print(f"unfused (LN1, QKV GEMM, attention, proj GEMM, LN2): {u:.3f} ms", flush=True)

if "--stamps" in sys.argv:
    import ctypes
    lib = L.load()
    names = ["barrier wait", "QKV", "pack/save", "S+softmax", "PV+proj", "stage store", "prologue", "epilogue"]
    for label, fn in (("inference", lambda: ops.wmsa_block_fwd(x, wpk, xm, xn2, None, None, None, None, None, None, B, H, H, C, HEADS, WS, 0)),
                      ("training", lambda: ops.wmsa_block_fwd(x, wpk, xm, xn2, st1, st2, xn1, None, lsew, ao, B, H, H, C, HEADS, WS, 0))):
        lib.sodt_debug_wmsa_stamps(None, 1)
        fn(); torch.cuda.synchronize(); big.zero_(); fn(); torch.cuda.synchronize()
        buf = (ctypes.c_longlong * (256 * 8))()
        lib.sodt_debug_wmsa_stamps(buf, 0)
        t = torch.tensor(list(buf), dtype=torch.float64).view(256, 8)
        tot = t.sum(1).mean()
        print(f"stamps ({label}; mean over 256 workgroups, wave 0; cycles per launch {tot:.0f}):")
        for i, n in enumerate(names):
            per = t[:, i].mean() / (M / 64 / 1024 * (12 if i < 6 else 1))
            print(f"   {n:14s} {t[:, i].mean():12.0f} cycles = {100 * t[:, i].mean() / tot:5.1f} %   ({per:8.0f} per {'head' if i < 6 else 'window'})")

if "--hg-stamps" in sys.argv:
    import ctypes
    lib = L.load()
    names = ["LN1", "B1 wait", "QKV", "B2/4/6 wait", "dma issue+saves", "softmax+PV", "B3/5 wait", "O^T->tile + B7", "projection",
             "B8 wait", "resid+staging+B9", "epilogue"]
    for label, fn in (("inference", lambda: ops.wmsa_block_fwd(x, wpk, xm, xn2, None, None, None, None, None, None, B, H, H, C, HEADS, WS, 0)),
                      ("training", lambda: ops.wmsa_block_fwd(x, wpk, xm, xn2, st1, st2, xn1, None, lsew, ao, B, H, H, C, HEADS, WS, 0))):
        lib.sodt_debug_wmsa_hg_stamps(None, 1)
        fn(); torch.cuda.synchronize(); big.zero_(); fn(); torch.cuda.synchronize()
        buf = (ctypes.c_longlong * (512 * 12))()
        lib.sodt_debug_wmsa_hg_stamps(buf, 0)
        t2 = torch.tensor(list(buf), dtype=torch.float64).view(2, 256, 12)
        npair = M / 64 / 2 / 256
        tot = t2[0].sum(1).mean()
        print(f"hg stamps ({label}; mean over 256 workgroups; cycles per launch {tot:.0f}, {tot / npair:.0f} per window pair); per pair: wave 0 | wave 4")
        for i, n in enumerate(names):
            print(f"   {n:20s} {t2[0][:, i].mean() / npair:8.0f} ({100 * t2[0][:, i].mean() / tot:5.1f} %) | {t2[1][:, i].mean() / npair:8.0f}")

