import importlib, os, sys, ctypes, torch
sys.path.insert(0, "/root/repo")
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
L = importlib.import_module("small-object-detection-transformers_amd._lib")
dev = torch.device("cuda:0"); dt = torch.bfloat16
B, H, Cc, ws, heads = 8, 256, 192, 8, 12
M = B * H * H; L2 = 15
nwin = M // 64
qkvw = torch.randn(nwin, heads, 3, 64, 16, device=dev).to(dt); lsew = torch.randn(nwin, heads, 64, device=dev)
bt = torch.randn(heads, L2 * L2, device=dev) * 0.1
dout = torch.randn(M, Cc, device=dev).to(dt); dqkv = torch.empty(M, 3 * Cc, device=dev, dtype=dt); dbt = torch.zeros_like(bt)
f = lambda: ops.window_attn_bwd_wm(qkvw, bt, dout, lsew, dqkv, dbt, B, H, H, Cc, heads, ws, 0)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("attn_bwd_wm stage1: %.3f ms" % (e0.elapsed_time(e1) / 10))



