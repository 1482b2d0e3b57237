"""ONE launch of sodt_wmsa_block_bwd (recompute form) and ONE of sodt_window_attn_bwd_wm (saved q / k / v) at the bench shape, shift 2 -
the process the --pmc passes of tools/pmc_gemm.sh profile (`tools/pmc_gemm.sh r04 pmc_bwd`)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "small-object-detection-transformers_amd"
ops = importlib.import_module(PKG + ".ops"); L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0"); dt = torch.bfloat16
C, HEADS, WS, B, H = 192, 12, 8, 8, 256
M = B * H * H
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
qw, qb, pw, pb = r(3 * C, C, sc=0.1), r(3 * C, sc=0.1), r(C, C, sc=0.1), r(C, sc=0.1)
tab, n1w, n1b, n2w, n2b = r(225, HEADS, sc=0.3), 1 + r(C, sc=0.1), r(C, sc=0.1), 1 + r(C, sc=0.1), r(C, sc=0.1)
x = r(M, C).to(dt); dout = r(M, C).to(dt)
wpk = torch.zeros(ops.wmsa_pack_bytes(C, HEADS, WS, L.BF16) // 2, device=dev, dtype=dt)
ops.wmsa_pack(qw, qb, pw, pb, tab, n1w, n1b, n2w, n2b, wpk, C, HEADS, WS)
xm, xn2, xn1, ao = (torch.empty(M, C, device=dev, dtype=dt) for _ in range(4))
st1, st2 = torch.empty(M, 2, device=dev), torch.empty(M, 2, device=dev)
lsew = torch.empty(M // 64, HEADS, 64, device=dev)
bias_t = tab.t().contiguous()
dqkv = torch.empty(M, 3 * C, device=dev, dtype=dt); dbt = torch.zeros_like(bias_t)
qkvw = (torch.randn(M // 64, HEADS, 3, 64, 16, generator=g) * 0.5).to(dev).to(dt)
ops.wmsa_block_fwd(x, wpk, xm, xn2, st1, st2, xn1, None, lsew, ao, B, H, H, C, HEADS, WS, 2)
torch.cuda.synchronize()
ops.wmsa_block_bwd(xn1, wpk, bias_t, dout, lsew, dqkv, dbt, B, H, H, C, HEADS, WS, 2)
torch.cuda.synchronize()
ops.window_attn_bwd_wm(qkvw, bias_t, dout, lsew, dqkv, dbt, B, H, H, C, HEADS, WS, 2)
torch.cuda.synchronize()
print("done", flush=True)
