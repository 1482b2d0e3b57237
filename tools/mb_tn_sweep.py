"""Fixed-cost vs streaming-cost of the TN kernels: M sweep at fixed splits."""
import importlib, os, sys
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
big = torch.empty(1 << 28, device=dev, dtype=torch.float32)   # 1 GiB: flush L2 / MALL between timed calls
def run(M, N, K, splits, v=0):
    dY = torch.randn(M, N, device=dev).to(dt); X = torch.randn(M, K, device=dev).to(dt)
    dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    ops.gemm_set_variant(v)
    def f():
        big.zero_()
        ops.gemm_tn(dY, [ops.SegSpec(X)], dW, M, N, K, dbias=db, splits=splits)
    tz = timeit(lambda: big.zero_())
    t = timeit(f) - tz
    ops.gemm_set_variant(0)
    print(f"v={v} M={M:7d} N={N} K={K} splits={splits:4d}: {t*1e3:8.1f} us  ({(M*(N+K)*2)/t/1e6:6.0f} GB/s)", flush=True)
for M in (65536, 131072, 262144, 524288):
    run(M, 192, 192, 256)
for sp in (64, 128, 256, 512, 1024):
    run(524288, 192, 192, sp)
run(524288, 192, 192, 256, v=2)
for sp in (85, 170, 255, 510):
    run(524288, 576, 192, sp)
run(524288, 576, 192, 85, v=2)
