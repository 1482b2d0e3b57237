"""Per-launch timing of one training step (HIP events around every recorded launch of the plan)."""
import importlib, os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    B, S = int(os.environ.get("B", 8)), int(os.environ.get("S", 1024))
    dev = torch.device("cuda:0")
    model = bench.build_model(S, dev, torch.bfloat16)
    x = torch.rand(B, 3, S, S, device=dev); ir = torch.rand(B, 3, S, S, device=dev)
    def step():
        pred, _ = model(x, ir, "RGB+IR"); pred[0].float().square().mean().backward()
        for p in model.parameters(): p.grad = None
    for _ in range(3): step()
    eng = model._get_engine(); plan = eng.plans[(B, S, torch.bfloat16, True)]
    res = collections.OrderedDict()
    for which, calls in (("fwd", plan.fwd_main), ("bwd", plan.bwd_main)):
        evs = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in range(len(calls))}
        setattr(eng, "probes_" + which, evs)
        step(); torch.cuda.synchronize()
        setattr(eng, "probes_" + which, None)
        for i, c in enumerate(calls):
            res[(which, i)] = (c[3], c[2], evs[i][0].elapsed_time(evs[i][1]))
    tot = sum(v[2] for v in res.values())
    print(f"sum of launches {tot:.2f} ms")
    agg = collections.defaultdict(float)
    for (which, i), (tag, name, ms) in res.items():
        stage = tag.split(".")[0] + (".bwd" if tag.endswith(".bwd") else "")
        agg[(which, stage, name)] += ms
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{k[0]:4s} {k[1]:16s} {k[2]:28s} {v:8.3f} ms")
    print("---- one stage-1 linear block (stage1.0) and one conv block (stage1.1), launch by launch")
    for (which, i), (tag, name, ms) in res.items():
        if tag in ("stage1.0", "stage1.0.bwd", "stage1.1", "stage1.1.bwd"):
            print(f"{which} {tag:14s} {name:26s} {ms:7.3f}")
main()
