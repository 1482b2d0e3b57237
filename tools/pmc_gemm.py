"""ONE launch each of the pipelined bf16 NT GEMM at four bench shapes - the process rocprofv3 --pmc passes profile
(tools/pmc_gemm.sh).  Order of the gemm_nt3 dispatches: (a) M=524288 N=768 K=384 bias+GELU (2x2-conv MLP of stage 1, flags 513),
(b) M=524288 N=192 K=768 plain, (c) M=131072 N=384 K=1536 plain (stage 2), (d) M=524288 N=192 K=576 plain."""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("small-object-detection-transformers_amd.ops")
dev = torch.device("cuda:0"); dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
SHAPES = [(524288, 768, 384, "gelu"), (524288, 192, 768, "plain"), (131072, 384, 1536, "plain"), (524288, 192, 576, "plain")]
calls = []
for M, N, K, mode in SHAPES:
    A = torch.randn(M, K, generator=g).to(dev).to(dt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    kw = dict(bias=torch.randn(N, generator=g).to(dev), gelu_only=True) if mode == "gelu" else {}
    calls.append((A, W, out, M, N, K, kw))
torch.cuda.synchronize()
for A, W, out, M, N, K, kw in calls:
    ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, **kw)
    torch.cuda.synchronize()
print("done", flush=True)
