"""End-to-end sanity on the GPU box: (a) 30 torch-SGD steps on a fixed synthetic batch with loss = mean(pred^2) must fall by
> 2x; (b) 40 steps of the real training step of bench.py (YOLOv5 ComputeLoss on fixed synthetic boxes, FusedSGD with the
weight-decay groups, ModelEMA) at 512^2 bf16: the loss must fall and stay finite, and the EMA model must evaluate."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
PKG = bench.PKG
dev = torch.device("cuda:0")
model = bench.build_model(512, dev, torch.bfloat16)
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.937, nesterov=True)
x = torch.rand(4, 3, 512, 512, device=dev); ir = torch.rand(4, 3, 512, 512, device=dev)
ls = []
for i in range(30):
    pred, _ = model(x, ir, "RGB+IR")
    loss = pred[0].float().square().mean()
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
    ls.append(float(loss))
print("mse :", " ".join(f"{v:.4f}" for v in ls[::3]))
assert ls[-1] < 0.5 * ls[0] and all(v == v for v in ls)

O = importlib.import_module(PKG + ".optim"); LS = importlib.import_module(PKG + ".loss")
model = bench.build_model(512, dev, torch.bfloat16)
model.hyp, model.gr, model.nc = dict(LS.DEFAULT_HYP), 1.0, 8
ema = O.ModelEMA(model)
opt = O.FusedSGD(O.set_weight_decay(model), model=model, lr=0.01, momentum=0.937, nesterov=True, ema=ema)
compute_loss = LS.ComputeLoss(model)
targets = LS.synthetic_targets(4, 32, 8, seed=0).to(dev)
ls = []
for i in range(40):
    pred, _ = model(x, ir, "RGB+IR")
    loss, lbox, lobj, lcls = compute_loss(pred, targets)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True); ema.update(model)
    ls.append(float(loss) / 4)
print("yolo:", " ".join(f"{v:.4f}" for v in ls[::4]), "| last components box/obj/cls", float(lbox), float(lobj), float(lcls))
assert all(v == v for v in ls) and ls[-1] < 0.8 * ls[0]
ema.ema.eval()
with torch.no_grad():
    z = ema.ema(x, ir, "RGB+IR")[0]
assert torch.isfinite(z).all()
print("ok")
