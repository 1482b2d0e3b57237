"""Per-wave cycle shares of the fused MLP kernel's steady-state steps, from a -DML_STAMPS build (tools/exp/ab_build.sh mlp "-DML_STAMPS" stamps).
usage: SODT_LIB_PATH=small-object-detection-transformers_amd/libsodt_hip_stamps.so python tools/mb_mlp_stamps.py [--save]"""
import ctypes as C
import importlib
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
L = importlib.import_module("small-object-detection-transformers_amd._lib")
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
Cc, M, dt = 192, 8 * 256 * 256, torch.bfloat16
g = torch.Generator().manual_seed(0)
xn = torch.randn(M, Cc, generator=g).to(dev).to(dt)
resid = torch.randn(M, Cc, generator=g).to(dev).to(dt)
w1 = (torch.randn(4 * Cc, Cc, generator=g) / math.sqrt(Cc)).to(dev).to(dt)
w2 = (torch.randn(Cc, 4 * Cc, generator=g) / math.sqrt(4 * Cc)).to(dev).to(dt)
b1 = torch.randn(4 * Cc, generator=g).to(dev)
b2 = torch.randn(Cc, generator=g).to(dev)
out = torch.empty(M, Cc, device=dev, dtype=dt)
hact = torch.empty(M, 4 * Cc, device=dev, dtype=dt) if "--save" in sys.argv else None
for _ in range(3):
    ops.mlp_fwd(xn, w1, b1, w2, b2, resid, out, hact, M, Cc)
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
rc = lib.sodt_debug_mlp_stamps(buf.ctypes.data_as(C.c_void_p))
assert rc == 0
a = buf.reshape(256, 8, 8).astype(np.float64)
nstep = 8 * 22          # 8 tiles per workgroup, 22 steady-state steps per tile
names = ["vmcnt wait", "barrier", "DMA issue", "step body"]
print(f"cycles per steady-state step (mean over 256 workgroups; {'training' if hact is not None else 'inference'} form)")
print("wave " + " ".join(f"{n:>11s}" for n in names) + "       total")
for w in range(8):
    v = a[:, w, :4].mean(0) / nstep
    print(f"  {w}  " + " ".join(f"{x:11.0f}" for x in v) + f" {v.sum():11.0f}")
