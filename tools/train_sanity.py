"""End-to-end sanity on the GPU box: 30 SGD steps on a fixed synthetic batch, loss = mean(pred^2) must fall by > 2x."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
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
print(" ".join(f"{v:.4f}" for v in ls[::3]))
assert ls[-1] < 0.5 * ls[0] and all(v == v for v in ls)
print("ok")
