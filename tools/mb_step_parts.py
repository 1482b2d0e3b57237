"""Where the bench step goes outside the engine's launch plans: optimizer, loss, zero_grad."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
model = bench.build_model(1024, dev, torch.bfloat16)
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.937, nesterov=True)
x = torch.rand(8, 3, 1024, 1024, device=dev); ir = torch.rand(8, 3, 1024, 1024, device=dev)
def run(do_opt, n=10):
    for _ in range(3):
        pred, _ = model(x, ir, "RGB+IR"); pred[0].float().square().mean().backward()
        if do_opt: opt.step()
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        pred, _ = model(x, ir, "RGB+IR"); pred[0].float().square().mean().backward()
        if do_opt: opt.step()
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"with optimizer {run(True):.2f} ms/step, without {run(False):.2f} ms/step")
