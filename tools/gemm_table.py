"""Every GEMM launch of one training step with its shape, time, TFLOP/s and the time its own HBM / MFMA bound allows
(bytes at 6 TB/s, flops at 1.6 PFLOP/s - practical ceilings), sorted by the time above that bound."""
import importlib, os, sys, collections, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
L = importlib.import_module(bench.PKG + "._lib")

def unwrap(a):
    return a._obj if hasattr(a, "_obj") else a

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
    rows = []
    other = collections.defaultdict(float)
    for which, calls in (("fwd", plan.fwd_main), ("bwd", plan.bwd_main)):
        evs = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in range(len(calls))}
        acc = collections.defaultdict(float)
        for rep in range(3):
            setattr(eng, "probes_" + which, evs)
            step(); torch.cuda.synchronize()
            setattr(eng, "probes_" + which, None)
            for i in range(len(calls)):
                acc[i] += evs[i][0].elapsed_time(evs[i][1]) / 3
        for i, c in enumerate(calls):
            fn, args, name, tag = c
            ms = acc[i]
            if name == "sodt_gemm_nt":
                g = unwrap(args[0])
                ptrs = set(g.a.s[k].p for k in range(g.a.nseg))
                # unique input columns: one klen per distinct (pointer) when taps re-read the same tensor
                kin = 0; seen = set()
                for k in range(g.a.nseg):
                    s = g.a.s[k]
                    key = s.p
                    if g.a.spatial and key in seen: continue
                    seen.add(key); kin += s.klen / ((s.mul * s.mul) if (g.a.spatial and s.shr) else 1)
                nout = 2 if g.flags & 4 else 1
                extra = (1 if g.flags & 2 else 0) + (1 if g.flags & 8 else 0)
                byt = 2.0 * g.M * (kin + g.N * (nout + extra))
                fl = 2.0 * g.M * g.N * g.K
                rows.append((which, tag, "nt", g.M, g.N, g.K, g.flags, ms, fl, byt))
            elif name == "sodt_gemm_tn":
                g = unwrap(args[0])
                kin = 0; seen = set()
                for k in range(g.x.nseg):
                    s = g.x.s[k]
                    if g.x.spatial and s.p in seen: continue
                    seen.add(s.p); kin += s.klen
                byt = 2.0 * g.M * (kin + g.N)
                fl = 2.0 * g.M * g.N * g.K
                rows.append((which, tag, "tn", g.M, g.N, g.K, g.splits, ms, fl, byt))
            else:
                other[name] += ms
    tot = sum(r[7] for r in rows)
    print(f"GEMM launches: {len(rows)}, {tot:.2f} ms; non-GEMM {sum(other.values()):.2f} ms")
    def bound(r): return max(r[9] / 6e12, r[8] / 1.6e15) * 1e3
    print(f"sum of GEMM bounds {sum(bound(r) for r in rows):.2f} ms")
    agg = collections.OrderedDict()
    for r in rows:
        k = (r[2], r[3], r[4], r[5], r[6])
        a = agg.setdefault(k, [0, 0.0, 0.0, r[8], r[9], set()])
        a[0] += 1; a[1] += r[7]; a[2] += bound(r); a[5].add(r[1].split(".")[0])
    print("kind        M      N      K  flg   n   total ms  bound ms  excess   TF/s   GB/s(alg)  stages")
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2])):
        t = a[1] / a[0]
        print(f"{k[0]:3s} {k[1]:8d} {k[2]:6d} {k[3]:6d} {k[4]:4d} {a[0]:3d} {a[1]:9.3f} {a[2]:9.3f} {a[1]-a[2]:7.3f} {a[3]/t/1e9:7.0f} {a[4]/t/1e6:8.0f}   {','.join(sorted(a[5]))}")
    for k, v in sorted(other.items(), key=lambda kv: -kv[1]):
        print(f"other {k:30s} {v:8.3f}")
main()
