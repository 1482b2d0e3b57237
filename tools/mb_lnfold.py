"""The input-gradient GEMM with the LayerNorm backward as epilogue (SODT_EPI_LNBWD) against the two launches it replaces, at the
stage-1 bench shapes (M = 524,288, N = 192), L2 / Infinity-Cache flush between calls."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "small-object-detection-transformers_amd"
ops = importlib.import_module(PKG + ".ops")
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
M, N = 524288, 192
g = torch.Generator().manual_seed(0)
x = torch.randn(M, N, generator=g).to(dev).to(dt); res = torch.randn(M, N, generator=g).to(dev).to(dt)
gam = torch.ones(N, device=dev); st = torch.zeros(M, 2, device=dev); y = torch.empty_like(x)
ops.layernorm_fwd(x, gam, torch.zeros(N, device=dev), y, st, M, N)
dx = torch.empty_like(x); dy = torch.empty_like(x); dg = torch.zeros(N, device=dev); db = torch.zeros(N, device=dev)
for K in (192, 576, 768):
    a = torch.randn(M, K, generator=g).to(dev).to(dt); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dt)
    def fused(): big.zero_(); ops.gemm_nt([ops.SegSpec(a)], w, dx, M, N, K, resid=res, ln_bwd=(x, st, gam, dg, db))
    def two(): big.zero_(); ops.gemm_nt([ops.SegSpec(a)], w, dy, M, N, K); ops.layernorm_bwd(dy, x, st, gam, res, dx, dg, db, M, N)
    def gemm_only(): big.zero_(); ops.gemm_nt([ops.SegSpec(a)], w, dy, M, N, K)
    f, t2, go = timeit(fused) - tz, timeit(two) - tz, timeit(gemm_only) - tz
    mb = M * (K + 3 * N) * 2 / 1e6
    print(f"K={K}: fused {f:.3f} ms ({mb / f / 1e3:.0f} GB/s of its {mb:.0f} MB), GEMM + layernorm_bwd {t2:.3f} ms (GEMM alone {go:.3f})", flush=True)
