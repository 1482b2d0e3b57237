"""Per-launch table of the super-resolution branch alone (DeepLab(4, 128, 512): Decoder + EDSR x8) at BASELINE config 5's grid:
B=4 @2048^2 -> the stride-4 tap is 4 x 512 x 512 rows, the x8 tail ends at 67 M rows.  Every live launch of SRBranch.forward /
.backward is timed with HIP events (one synchronise per launch: a table, not a step time).

  python tools/sr_launch_table.py [out.md]      (env B=4, H=512)
"""
import collections, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import roofline_table as RT
PKG = "small-object-detection-transformers_amd"
ops = importlib.import_module(PKG + ".ops"); sr = importlib.import_module(PKG + ".sr")


def main():
    B, H = int(os.environ.get("B", 4)), int(os.environ.get("H", 512))
    dev = torch.device("cuda:0"); dt = torch.bfloat16
    torch.manual_seed(0)
    m = sr.DeepLab(4, 128, 512).to(dev)
    params = {k: v.detach() for k, v in m.state_dict().items()}
    br = sr.SRBranch(params, dt)
    low = (torch.randn(B * H * H, 128, device=dev) * 0.5).to(dt)
    x = (torch.randn(B * (H // 2) ** 2, 512, device=dev) * 0.5).to(dt)
    dy = torch.randn(B, 4, 8 * H, 8 * H, device=dev) * 1e-3

    def step():
        y = br.forward([ops.SegSpec(low)], [ops.SegSpec(x)], B, H, H)
        br.backward(dy)
        return y
    step(); torch.cuda.synchronize()
    rec = []
    real = ops._launch

    def timed(name, *args):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); real(name, *args); e.record(); e.synchronize()
        shape, byt, fl = RT.cost(name, args)
        rec.append((name.replace("sodt_", ""), shape, s.elapsed_time(e), byt, fl))
    ops._launch = timed
    step()
    ops._launch = real
    tot = sum(r[2] for r in rec)
    groups = collections.OrderedDict()
    for name, shape, ms, byt, fl in rec:
        g = groups.setdefault((name, shape), [0, 0.0, 0.0, 0.0])
        g[0] += 1; g[1] += ms; g[2] += byt; g[3] += fl
    lines = [f"# SR branch alone, B={B}, tap grid {H}x{H} (bf16): every live launch, HIP events (tools/sr_launch_table.py)", "",
             f"Sum: {tot:.1f} ms (forward + backward), peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB.", "",
             "| entry point | shape | launches | ms | % | alg. MB / launch | GFLOP / launch | GB/s | TFLOP/s |", "|---|---|---|---|---|---|---|---|---|"]
    for (name, shape), (n, ms, byt, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {name} | {shape} | {n} | {ms:.2f} | {100 * ms / tot:.1f} | {byt / n / 1e6:.0f} | {fl / n / 1e9:.1f} | "
                     f"{byt / ms / 1e6 if ms else 0:.0f} | {fl / ms / 1e9 if ms else 0:.0f} |")
    out = "\n".join(lines) + "\n"
    print(out)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(out)


if __name__ == "__main__":
    main()
