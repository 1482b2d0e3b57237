import importlib, os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
from tools.microbench import timeit
dev = torch.device("cuda:0"); dt = torch.bfloat16
M, N, K = 8 * 256 * 256, 768, 192
A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
out = torch.empty(M, N, device=dev, dtype=dt)
for name, fl in (("full", 0), ("nostore", 1 << 20), ("nomfma", 1 << 21), ("noW", 1 << 22), ("nostore+nomfma", 3 << 20), ("only A load", 7 << 20)):
    ms = timeit(lambda: ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, debug_flags=fl))
    print(f"{name:16s} {ms:.3f} ms")
# raw copy bandwidth reference
x = torch.empty(M * N, device=dev, dtype=dt); y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x)); print(f"torch copy {2*x.numel()*2/ms/1e6:.0f} GB/s")
ms = timeit(lambda: y.zero_()); print(f"torch fill {x.numel()*2/ms/1e6:.0f} GB/s")
