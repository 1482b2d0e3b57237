"""ONE inference-form and ONE training-form launch of the fused W-MSA block kernel at the bench shape (B=8 @1024^2: T = 524,288
tokens, C = 192) - the process rocprofv3 --pmc passes profile (tools/pmc_mem.sh).  A counter pass replays every launch of the
process once per counter group, so this driver keeps the launch count minimal (the TA/TCP pass over
tools/mb_wmsa.py's timing loops did not finish)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "small-object-detection-transformers_amd"
ops = importlib.import_module(PKG + ".ops")
L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0"); dt = torch.bfloat16
C, HEADS, WS, B, H = 192, 12, 8, 8, 256
M = B * H * H
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
qw, qb, pw, pb = r(3 * C, C, sc=0.1), r(3 * C, sc=0.1), r(C, C, sc=0.1), r(C, sc=0.1)
tab, n1w, n1b, n2w, n2b = r(225, HEADS, sc=0.3), 1 + r(C, sc=0.1), r(C, sc=0.1), 1 + r(C, sc=0.1), r(C, sc=0.1)
x = r(M, C).to(dt)
wpk = torch.zeros(ops.wmsa_pack_bytes(C, HEADS, WS, L.BF16) // 2, device=dev, dtype=dt)
ops.wmsa_pack(qw, qb, pw, pb, tab, n1w, n1b, n2w, n2b, wpk, C, HEADS, WS)
xm, xn2, xn1, ao = (torch.empty(M, C, device=dev, dtype=dt) for _ in range(4))
st1, st2 = torch.empty(M, 2, device=dev), torch.empty(M, 2, device=dev)
lsew = torch.empty(M // 64, HEADS, 64, device=dev)
torch.cuda.synchronize()
shift = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ops.wmsa_block_fwd(x, wpk, xm, xn2, None, None, None, None, None, None, B, H, H, C, HEADS, WS, shift)
torch.cuda.synchronize()
ops.wmsa_block_fwd(x, wpk, xm, xn2, st1, st2, xn1, None, lsew, ao, B, H, H, C, HEADS, WS, shift)
torch.cuda.synchronize()
print("done", flush=True)
