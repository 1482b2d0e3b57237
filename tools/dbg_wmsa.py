import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_wmsa_block_gpu as T
PKG = "small-object-detection-transformers_amd"
ops = importlib.import_module(PKG + ".ops"); L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")
C, HEADS, WS, HD = T.C, T.HEADS, T.WS, T.HD
for dt in (torch.float32, torch.bfloat16):
    for (B, H, W, shift) in [(2, 16, 16, 0), (2, 16, 24, 2), (3, 8, 8, 0)]:
        sd = T._params(dev, seed=1)
        M = B * H * W
        x = (torch.randn(M, C, generator=torch.Generator().manual_seed(7)) * 1.3 + 0.2).to(dev).to(dt)
        sdr = {k: (v.to(dt).float() if v.dim() == 2 and "table" not in k else v) for k, v in sd.items()} if dt == torch.bfloat16 else sd
        ref = T._reference(sdr, x.float(), B, H, W, shift)
        wpk = T._pack(ops, L, sd, dev, dt)
        nwin = M // 64
        outs = dict(xm=torch.full((M, C), 7.0, device=dev, dtype=dt), xn2=torch.full((M, C), 7.0, device=dev, dtype=dt),
                    st1=torch.zeros(M, 2, device=dev), st2=torch.zeros(M, 2, device=dev),
                    xn1=torch.zeros(M, C, device=dev, dtype=dt), qkvw=torch.zeros(nwin, HEADS, 3, 64, HD, device=dev, dtype=dt),
                    lse=torch.zeros(nwin, HEADS, 64, device=dev), ao=torch.zeros(M, C, device=dev, dtype=dt))
        ops.wmsa_block_fwd(x, wpk, outs["xm"], outs["xn2"], outs["st1"], outs["st2"], outs["xn1"], outs["qkvw"], outs["lse"],
                           outs["ao"], B, H, W, C, HEADS, WS, shift)
        torch.cuda.synchronize()
        line = f"{str(dt):16s} B{B} {H}x{W} s{shift}: "
        for name in ("st1", "xn1", "qkvw", "lse", "ao", "xm"):
            e, s = T._err(outs[name], ref[name])
            line += f"{name} {e:.2e}/{s:.1e}  "
        q = outs["qkvw"].float().cpu().double(); rq = ref["qkvw"]
        for i, nm in enumerate("qkv"):
            line += f"{nm} {float((q[:, :, i] - rq[:, :, i]).abs().max()):.2e} "
        print(line, flush=True)
