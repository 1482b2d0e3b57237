import sys, os, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_model_gpu import build
from test_bf16_parity_gpu import _train_step, _compare, GOLD
from oracle import ref_torch as R
dev = torch.device("cuda:0")
gold = torch.load(os.path.join(GOLD, "autocast_512.pt"))
model, _ = build(dev, 512)
x_rgb, x_ir = R.synthetic_inputs(1, 512, seed=0)
x_rgb, x_ir = x_rgb.to(dev), x_ir.to(dev)
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
lf, gf = _train_step(model, torch.float32, x_rgb, x_ir)
model.load_state_dict(sd0)
lb, gb = _train_step(model, torch.bfloat16, x_rgb, x_ir)
gmed = sorted(float(g.norm()) for g in gf.values())[len(gf) // 4]
dmax, dmean, grel = _compare(lf, gf, lb, gb, 1e-2 * gmed)
rat = sorted(((v / max(gold["grad_rel"][k], 1e-9), k, v, gold["grad_rel"][k]) for k, v in grel.items() if k != "image_encoder.stage3.0.mlp.fc2.bias"), reverse=True)
import statistics
r = [x[0] for x in rat]
print("logits", dmax, dmean, gold["logit_maxdiff"], gold["logit_meandiff"])
print("ratio quantiles: max %.3f p95 %.3f median %.3f mean %.3f min %.3f" % (r[0], r[len(r)//20], statistics.median(r), sum(r)/len(r), r[-1]))
for x in rat[:8]: print("%.3f %s ours %.4f ref %.4f" % x)
