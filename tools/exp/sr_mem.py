import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
PKG = "small-object-detection-transformers_amd"
ops = importlib.import_module(PKG + ".ops"); sr = importlib.import_module(PKG + ".sr")
dev = torch.device("cuda:0"); dt = torch.bfloat16
B, H = 4, 512
m = sr.DeepLab(4, 128, 512).to(dev)
br = sr.SRBranch({k: v.detach() for k, v in m.state_dict().items()}, dt)
low = (torch.randn(B * H * H, 128, device=dev) * 0.5).to(dt); x = (torch.randn(B * (H // 2) ** 2, 512, device=dev) * 0.5).to(dt)
dy = torch.randn(B, 4, 8 * H, 8 * H, device=dev) * 1e-3
y = br.forward([ops.SegSpec(low)], [ops.SegSpec(x)], B, H, H); br.backward(dy); torch.cuda.synchronize()
tot = 0
for k, v in sorted(br.bufs.items(), key=lambda kv: -kv[1].numel() * kv[1].element_size())[:18]:
    print(f"{k:14s} {tuple(v.shape)} {v.numel() * v.element_size() / 2**30:.2f} GiB")
print("sum of bufs", sum(v.numel() * v.element_size() for v in br.bufs.values()) / 2**30, "GiB; peak", torch.cuda.max_memory_allocated() / 2**30)
