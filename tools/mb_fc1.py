import importlib, os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
dev = torch.device("cuda:0"); dt = torch.bfloat16
big = torch.empty(1 << 28, device=dev, dtype=torch.float32)
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
tz = timeit(lambda: big.zero_())
M, N, K = 524288, 768, 192
A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
out = torch.empty(M, N, device=dev, dtype=dt); bias = torch.randn(N, device=dev)
for name, kw in (("plain", {}), ("bias", dict(bias=bias)), ("bias+gelu", dict(bias=bias, gelu_only=True))):
    def f(): big.zero_(); ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, **kw)
    t = timeit(f) - tz
    print(f"{name:10s} {t*1e3:7.1f} us  {(M*(K+N)*2)/t/1e6:6.0f} GB/s")
for (N2, K2) in ((192, 768), (192, 192), (576, 192), (384, 192), (1536, 192)):
    A2 = torch.randn(M, K2, device=dev).to(dt); W2 = (torch.randn(N2, K2, device=dev) / math.sqrt(K2)).to(dt); o2 = torch.empty(M, N2, device=dev, dtype=dt)
    def f(): big.zero_(); ops.gemm_nt([ops.SegSpec(A2)], W2, o2, M, N2, K2)
    t = timeit(f) - tz
    print(f"N={N2} K={K2} plain {t*1e3:7.1f} us  {(M*(K2+N2)*2)/t/1e6:6.0f} GB/s")
