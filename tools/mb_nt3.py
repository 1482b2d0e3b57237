"""A/B of the pipelined bf16 NT kernel (variant 0 = auto) against the 128x128 K-loop kernel (variant 1) at the bench shapes."""
import importlib, os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
dev = torch.device("cuda:0"); dt = torch.bfloat16

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def run(M, N, K, mode, taps=False):
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt); bias = torch.randn(N, device=dev)
    kw = {}
    if taps:   # 2x2 conv taps over a (B, H, H, K/4) image
        C = K // 4; H = int(math.isqrt(M // 8))
        x = torch.randn(M, C, device=dev).to(dt)
        segs = [ops.SegSpec(x, C, 0, dy, dx, 1, 0, H, H) for dy in (0, 1) for dx in (0, 1)]
        kw["spatial"] = (H, H)
    else:
        A = torch.randn(M, K, device=dev).to(dt)
        segs = [ops.SegSpec(A)]
    if mode in ("bias", "resid", "gelu"): kw["bias"] = bias
    if mode == "resid": kw["resid"] = torch.randn(M, N, device=dev).to(dt)
    if mode == "gelu": kw["gelu_out"] = torch.empty(M, N, device=dev, dtype=dt)
    if mode == "dgelu": kw["dgelu_aux"] = torch.randn(M, N, device=dev).to(dt)
    res = []
    for rep in range(2):
        for v in (3, 0):
            ops.gemm_set_variant(v)
            res.append((v, timeit(lambda: ops.gemm_nt(segs, W, out, M, N, K, **kw))))
    ops.gemm_set_variant(0)
    t0 = min(t for v, t in res if v == 3); t1 = min(t for v, t in res if v == 0)
    print(f"M={M:7d} N={N:5d} K={K:5d} {mode:6s} taps={int(taps)}: pipelined {t0:7.3f} ms {2*M*N*K/t0/1e9:7.0f} TF/s | auto {t1:7.3f} ms {2*M*N*K/t1/1e9:7.0f} TF/s | x{t1/t0:.2f}", flush=True)

import sys
shapes = [(524288, 192, 192, "plain"), (524288, 576, 192, "bias"), (524288, 768, 192, "gelu"), (524288, 768, 192, "dgelu"),
          (524288, 192, 192, "resid"), (524288, 192, 192, "bias"), (131072, 384, 384, "resid"), (131072, 1152, 384, "bias")]
for a in shapes:
    run(*a)
