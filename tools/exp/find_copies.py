"""Which host-side ops add small kernels / device copies to a training step?  (torch.profiler over ONE steady-state step)"""
import os, sys, collections, importlib
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity
PKG = bench.PKG
dev = torch.device("cuda:0")
B, S = 8, 1024
model = bench.build_model(S, dev, torch.bfloat16)
O = importlib.import_module(PKG + ".optim")
LS = importlib.import_module(PKG + ".loss")
ema = O.ModelEMA(model)
opt = O.FusedSGD(O.set_weight_decay(model), model=model, lr=0.01, momentum=0.937, nesterov=True, ema=ema)
x_rgb = torch.rand(B, 3, S, S).to(dev); x_ir = torch.rand(B, 3, S, S).to(dev)
model.hyp, model.gr, model.nc = dict(LS.DEFAULT_HYP), 1.0, 8
compute_loss = LS.ComputeLoss(model)
targets = LS.synthetic_targets(B, 32, 8, seed=0).to(dev)
def step():
    pred, _ = model(x_rgb, x_ir, "RGB+IR")
    loss = compute_loss(pred, targets)[0]
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True); ema.update(model)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = collections.Counter(); tm = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA or "hipMemcpy" in e.name or "hipMemset" in e.name:
        ev[e.name[:70]] += 1; tm[e.name[:70]] += e.cuda_time if hasattr(e, "cuda_time") else 0
for k, v in ev.most_common(30):
    if "sodt" in k or "GLOBAL__N" in k or "anonymous" in k: continue
    print(v, round(tm[k]), k)
st = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to") and e.stack:
        st[e.name + " @ " + " <- ".join(s.split("/")[-1] for s in e.stack[:4])] += 1
for k, v in st.most_common(20): print(v, k)
