"""Same-box A/B of engine switches on the bench workload (B=8 @1024^2 bf16, forward + backward, loss = mean(pred^2)): builds one model per
setting, interleaves rounds.  usage: python tools/ab_step_flags.py  (settings are listed below)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda:0")
SETTINGS = {
    "shipped": {},
    "no conv-MLP fold": {"convmlp_fold_maxc": 0},
    "no fused linear MLP": {"use_fused_mlp": False},
    "neither": {"convmlp_fold_maxc": 0, "use_fused_mlp": False},
    "no direct 3x3 in the head": {"use_direct_conv3": False},
}
if os.environ.get("AB_ONLY"):            # a comma-separated subset (each setting owns a 31 GB workspace)
    SETTINGS = {k: v for k, v in SETTINGS.items() if k in os.environ["AB_ONLY"].split(",")}
x = torch.rand(8, 3, 1024, 1024, device=dev)
ir = torch.rand(8, 3, 1024, 1024, device=dev)
models = {}
for name, kv in SETTINGS.items():
    m = bench.build_model(1024, dev, torch.bfloat16)
    eng = m._get_engine()
    for k, v in kv.items():
        assert hasattr(eng, k), k
        setattr(eng, k, v)
    models[name] = m


def run(m, n):
    for _ in range(n):
        pred, _ = m(x, ir, "RGB+IR")
        pred[0].float().square().mean().backward()
        m.zero_grad(set_to_none=True)


for m in models.values():
    run(m, 2)
torch.cuda.synchronize()
times = {k: [] for k in models}
for r in range(4):
    for k, m in models.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(m, 5)
        torch.cuda.synchronize()
        times[k].append((time.perf_counter() - t0) / 5 * 1e3)
for k, v in times.items():
    v = sorted(v)
    print(f"{k:24s} median {v[len(v) // 2]:.2f} ms  min {v[0]:.2f} ms   (forward + backward, no loss / optimizer kernels)")
