"""Stage-3 window attention (32x32 windows, 12 heads x 64, B=8 @ 1024^2 -> 64x64 tokens) fwd / bwd timings."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
dev = torch.device("cuda:0"); dt = torch.bfloat16
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def attn(B, H, Cc, ws, shift, heads=12):
    M = B * H * H; L2 = 2 * ws - 1
    qkv = torch.randn(M, 3 * Cc, device=dev).to(dt); bt = torch.randn(heads, L2 * L2, device=dev) * 0.1
    out = torch.empty(M, Cc, device=dev, dtype=dt); lse = torch.empty(M, heads, device=dev)
    ms = timeit(lambda: ops.window_attn_fwd(qkv, bt, out, lse, B, H, H, Cc, heads, ws, shift))
    dout = torch.randn(M, Cc, device=dev).to(dt); dqkv = torch.empty_like(qkv); dbt = torch.zeros_like(bt)
    scr = torch.zeros(M * (Cc + heads), device=dev)
    ms2 = timeit(lambda: ops.window_attn_bwd(qkv, bt, out, dout, lse, dqkv, dbt, scr, B, H, H, Cc, heads, ws, shift))
    fl = 4.0 * M * ws * ws * Cc
    print(f"attn C={Cc} H={H} ws={ws} shift={shift}: fwd {ms:7.3f} ms ({fl/ms/1e9:6.0f} TFLOP/s)  bwd {ms2:7.3f} ms ({2.5*fl/ms2/1e9:6.0f} TFLOP/s on 5 products)", flush=True)
attn(8, 64, 768, 32, 0)
attn(8, 64, 768, 32, 16)
attn(8, 32, 768, 16, 0)
