"""Pipelined (auto) vs 128x128 K-loop (tiled) NT GEMM on thin / tapped shapes: the head's and the SR branch's 3x3 convolutions."""
import importlib, math, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
dev = torch.device("cuda:0"); dt = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def run(M, N, K, taps):
    H = int(math.isqrt(M // 8)); M = 8 * H * H            # (a whole (8, H, H) image: the row -> pixel map must cover every row)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    kw = {}
    if taps:
        C = K // taps
        x = torch.randn(M, C, device=dev).to(dt)
        tl = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
        segs = [ops.SegSpec(x, C, 0, dy, dx, 1, 0, H, H) for dy, dx in tl]
        kw["spatial"] = (H, H)
    else:
        segs = [ops.SegSpec(torch.randn(M, K, device=dev).to(dt))]
    for v, name in ((0, "auto(nt3)"), (1, "tiled")):
        ops.gemm_set_variant(v)
        t = min(timeit(lambda: ops.gemm_nt(segs, W, out, M, N, K, **kw)) for _ in range(2))
        print(f"M={M} N={N} K={K} taps={taps} {name}: {t*1e3:.1f} us {2*M*N*K/t/1e9:.0f} TF/s", flush=True)
    ops.gemm_set_variant(0)
for a in [(524288, 64, 576, 9), (524288, 64, 576, 0), (524288, 192, 576, 9), (1048576, 256, 576, 9), (4194304, 64, 2304, 9)]:
    run(*a)
