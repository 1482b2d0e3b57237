"""ONE launch each of the pipelined bf16 TN (weight-gradient) GEMM at three stage-1 bench shapes - the process rocprofv3 --pmc
passes profile (tools/pmc_gemm.sh tn).  Order of the gemm_tn3 dispatches: (a) M=524288 N=192 K=192, (b) N=192 K=768, (c) N=576 K=192."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
dev = torch.device("cuda:0"); dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
scratch = torch.empty(16 << 20, dtype=torch.float32, device=dev)
ops.set_tn_scratch(scratch)
calls = []
for M, N, K in [(524288, 192, 192), (524288, 192, 768), (524288, 576, 192)]:
    dY = torch.randn(M, N, generator=g).to(dev).to(dt)
    X = torch.randn(M, K, generator=g).to(dev).to(dt)
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    calls.append((dY, X, dW, db, M, N, K))
torch.cuda.synchronize()
for dY, X, dW, db, M, N, K in calls:
    ops.gemm_tn(dY, [ops.SegSpec(X)], dW, M, N, K, dbias=db)
    torch.cuda.synchronize()
print("done", flush=True)
