"""Timing of the bf16 NT GEMM at the bench shapes with whatever library SODT_LIB_PATH names (A/B of two builds inside ONE gpurun
call: box-to-box spread is larger than most kernel changes).  Usage: [SODT_LIB_PATH=...] python tools/ab_gemm_lib.py [tag]"""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
dev = torch.device("cuda:0"); dt = torch.bfloat16
tag = sys.argv[1] if len(sys.argv) > 1 else "lib"


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def run(M, N, K, mode, taps=0):
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt); bias = torch.randn(N, device=dev)
    kw = {}
    if taps:   # 2x2 / 3x3 conv taps over a (8, H, H, K / taps) image
        C = K // taps; H = int(math.isqrt(M // 8))
        x = torch.randn(M, C, device=dev).to(dt)
        tl = [(dy, dx) for dy in (0, 1) for dx in (0, 1)] if taps == 4 else [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
        segs = [ops.SegSpec(x, C, 0, dy, dx, 1, 0, H, H) for dy, dx in tl]
        kw["spatial"] = (H, H)
    elif mode == "dgelu_rc":      # the Linear-GELU-Linear backward: A = [xn | dy], W = [W1 | W2^T], pre-activation recomputed in the first K half
        segs = [ops.SegSpec(torch.randn(M, K // 2, device=dev).to(dt)), ops.SegSpec(torch.randn(M, K // 2, device=dev).to(dt))]
        kw["bias"] = bias; kw["dgelu_rc"] = True
    else:
        segs = [ops.SegSpec(torch.randn(M, K, device=dev).to(dt))]
    if mode in ("bias", "gelu", "gelu2", "biasres"): kw["bias"] = bias
    if mode in ("resid", "biasres"): kw["resid"] = torch.randn(M, N, device=dev).to(dt)
    if mode == "gelu": kw["gelu_only"] = True
    if mode == "gelu2": kw["gelu_out"] = torch.empty(M, N, device=dev, dtype=dt)
    if mode == "dgelu": kw["dgelu_aux"] = torch.randn(M, N, device=dev).to(dt)
    t = min(timeit(lambda: ops.gemm_nt(segs, W, out, M, N, K, **kw)) for _ in range(2))
    print(f"[{tag}] M={M:7d} N={N:5d} K={K:5d} {mode:7s} taps={taps}: {t * 1e3:8.1f} us {2 * M * N * K / t / 1e9:6.0f} TF/s", flush=True)
    return t


shapes = [(524288, 768, 384, "dgelu_rc", 0), (131072, 1536, 768, "dgelu_rc", 0), (524288, 768, 384, "gelu", 0), (524288, 192, 768, "plain", 4), (524288, 192, 768, "plain", 0), (524288, 192, 576, "plain", 0),
          (524288, 192, 768, "biasres", 0), (524288, 192, 768, "dgelu", 0), (524288, 768, 192, "gelu2", 0), (524288, 192, 192, "plain", 0),
          (131072, 384, 1536, "plain", 0), (131072, 1536, 768, "gelu", 0), (131072, 1152, 384, "bias", 0), (131072, 384, 384, "biasres", 0),
          (32768, 3072, 1536, "gelu", 0), (32768, 768, 3072, "biasres", 0), (524288, 192, 1728, "plain", 9)]
print(f"[{tag}] {ops.version()}")
tot = sum(run(*a) for a in shapes)
print(f"[{tag}] sum {tot * 1e3:.1f} us", flush=True)
