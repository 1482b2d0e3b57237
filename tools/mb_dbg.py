import importlib, os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
from tools.microbench import timeit
dev = torch.device("cuda:0"); dt = torch.bfloat16
M, N, K = 8 * 256 * 256, 192, 768
A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt)
out = torch.empty(M, N, device=dev, dtype=dt)
for name, fl in (("full(spec)", 0), ("full(generic)", 1 << 25), ("nostore", 1 << 20), ("nomfma", 1 << 21), ("noload", 1 << 22), ("nostore+nomfma", 3 << 20), ("only loads", (1 << 20) | (1 << 21)), ("nothing", 7 << 20)):
    ms = timeit(lambda: ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, debug_flags=fl))
    print(f"{name:16s} {ms:.3f} ms")
