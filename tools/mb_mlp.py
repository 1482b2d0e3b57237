"""Same-box A/B of the fused linear MLP (sodt_mlp_fwd, csrc/mlp.hip) against the two-GEMM chain it replaces.
usage: python tools/mb_mlp.py [B] [--rounds N]   (stage-1 size: M = B * 256^2, C = 192)"""
import importlib
import math
import sys

import torch

sys.path.insert(0, ".")
ops = importlib.import_module("small-object-detection-transformers_amd.ops")

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 5
dev = torch.device("cuda:0")
Cc, M = 192, B * 256 * 256
dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
xn = torch.randn(M, Cc, generator=g).to(dev).to(dt)
resid = torch.randn(M, Cc, generator=g).to(dev).to(dt)
w1 = (torch.randn(4 * Cc, Cc, generator=g) / math.sqrt(Cc)).to(dev).to(dt)
w2 = (torch.randn(Cc, 4 * Cc, generator=g) / math.sqrt(4 * Cc)).to(dev).to(dt)
b1 = torch.randn(4 * Cc, generator=g).to(dev)
b2 = torch.randn(Cc, generator=g).to(dev)
out = torch.empty(M, Cc, device=dev, dtype=dt)
hact = torch.empty(M, 4 * Cc, device=dev, dtype=dt)
flush = torch.empty(160 * 1024 * 1024, device=dev, dtype=torch.float32)      # 640 MB: past L2 + Infinity Cache between variants


def chain():
    ops.gemm_nt([ops.SegSpec(xn)], w1, hact, M, 4 * Cc, Cc, bias=b1, gelu_only=True)
    ops.gemm_nt([ops.SegSpec(hact)], w2, out, M, Cc, 4 * Cc, bias=b2, resid=resid)


variants = {
    "two-GEMM chain (fc1 GELU-only + fc2 bias+resid)": chain,
    "fused, training form (stores GELU(h))": lambda: ops.mlp_fwd(xn, w1, b1, w2, b2, resid, out, hact, M, Cc),
    "fused, inference form": lambda: ops.mlp_fwd(xn, w1, b1, w2, b2, resid, out, None, M, Cc),
}
for f in variants.values():
    f()
torch.cuda.synchronize()
times = {k: [] for k in variants}
for r in range(rounds):
    for k, f in variants.items():
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            f()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 3)
flops = 16.0 * M * Cc * Cc
print(f"M={M} C={Cc} bf16, {flops / 1e9:.1f} GFLOP per call; median / min ms over {rounds} interleaved rounds")
for k, v in times.items():
    v = sorted(v)
    med = v[len(v) // 2]
    print(f"  {k:50s} {med:.3f} / {v[0]:.3f} ms   {flops / med / 1e9:.0f} TFLOP/s")
