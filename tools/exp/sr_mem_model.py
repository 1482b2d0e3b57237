"""Where the memory of Model(sr=True) at B=4 @2048^2 goes: allocated bytes after build / forward / backward, the engine plan's largest
buffers and the SR branch's."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
M = importlib.import_module("small-object-detection-transformers_amd.model")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, S = 4, 2048
x = torch.rand(B, 3, S, S, generator=g).to(dev); ir = torch.rand(B, 1, S, S, generator=g).to(dev)
gib = lambda: torch.cuda.memory_allocated() / 2**30
m = M.Model("SRyolo_MF.yaml", input_mode="RGB+IR", ch_steam=3, ch=128, nc=8, sr=True).to(dev)
m.compute_dtype = torch.bfloat16; m.train()
print(f"after build {gib():.1f} GiB")
out = m(x, ir, "RGB+IR"); torch.cuda.synchronize()
print(f"after forward {gib():.1f} GiB (peak {torch.cuda.max_memory_allocated() / 2**30:.1f})")
loss = out[0][0].float().square().mean() + out[1].square().mean()
loss.backward(); torch.cuda.synchronize()
print(f"after backward {gib():.1f} GiB (peak {torch.cuda.max_memory_allocated() / 2**30:.1f})")
eng = m._get_engine()
plan = next(iter(eng.plans.values()))
sz = lambda t: t.numel() * t.element_size() / 2**30
tot = sum(sz(t) for t in plan.bufs.values())
print(f"plan buffers: {len(plan.bufs)} tensors, {tot:.1f} GiB")
for k, v in sorted(plan.bufs.items(), key=lambda kv: -sz(kv[1]))[:14]:
    print(f"  {k:28s} {tuple(v.shape)} {sz(v):.2f} GiB")
for name in ("sr", "sr_branch", "srb"):
    br = getattr(plan, name, None) or getattr(eng, name, None)
    if br is not None and hasattr(br, "bufs"):
        print(f"SR branch buffers ({name}): {sum(sz(t) for t in br.bufs.values()):.1f} GiB")
