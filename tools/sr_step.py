"""Time the training step of Model(sr=True) (BASELINE config 5: SRyolo_MF.yaml, super-resolution branch, batch 4 @ 2048x2048)
next to the same step without the branch.  Usage: python tools/sr_step.py [--batch 4] [--size 2048] [--steps 5] [--dtype bf16]"""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    M = importlib.import_module("small-object-detection-transformers_amd.model")
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    g = torch.Generator().manual_seed(0)
    x = torch.rand(a.batch, 3, a.size, a.size, generator=g).to(dev)
    ir = torch.rand(a.batch, 1, a.size, a.size, generator=g).to(dev)
    for sr in (False, True):
        torch.manual_seed(0)
        m = M.Model("SRyolo_MF.yaml", input_mode="RGB+IR", ch_steam=3, ch=128, nc=8, sr=sr).to(dev)
        m.compute_dtype = dt
        m.train()

        def step():
            out = m(x, ir, "RGB+IR")
            loss = out[0][0].float().square().mean()
            if sr:
                loss = loss + out[1].square().mean()
            loss.backward()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(f"sr={sr}: B={a.batch} @ {a.size}^2 {a.dtype}: {ms:.1f} ms / step, {a.batch / ms * 1e3:.1f} img/s, "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
        del m, step
        import gc
        gc.collect()                      # (the engine and its recorded plans refer to each other: without a collection the first model's
        torch.cuda.empty_cache()          #  46 GiB workspace is still allocated while the second one runs and lands in its "peak")
        torch.cuda.reset_peak_memory_stats()
        print(f"  (still allocated after this model: {torch.cuda.memory_allocated() / 2**30:.1f} GiB)", flush=True)


if __name__ == "__main__":
    main()
