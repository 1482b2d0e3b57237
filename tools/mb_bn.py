"""sodt_bn_silu_bwd_reduce / _apply / _fwd at the head's shapes (h7: 512 K rows x 64 channels; h0 / h3: 128 K x 128), L2 flush between calls."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
for M, C in ((524288, 64), (524288, 128), (131072, 128), (131072, 256), (32768, 512)):
    z = torch.randn(M, C, device=dev).to(dt); dy = torch.randn(M, C, device=dev).to(dt)
    mr = torch.cat([torch.zeros(C), torch.ones(C)]).to(dev); ga = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
    red = torch.zeros(2, C, device=dev, dtype=torch.float64)
    def f(): big.zero_(); ops.bn_silu_bwd_reduce(dy, C, z, mr, ga, be, red, M, C)
    t = timeit(f) - tz
    print(f"bn_silu_bwd_reduce M={M} C={C}: {t * 1e3:.1f} us ({2 * M * C * 2 / t / 1e6:.0f} GB/s)", flush=True)
