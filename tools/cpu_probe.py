import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_torch as R
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
for th in (16, 8):
    torch.set_num_threads(th)
    sd = R.procedural_state_dict(256, 8)
    osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "anchor" not in k) for k, v in sd.items()}
    x_rgb, x_ir = R.synthetic_inputs(1, 256, seed=0)
    for i in range(2):
        t0 = time.perf_counter()
        pred, _ = R.model_forward(osd, x_rgb, x_ir, True, {})
        pred[0].square().mean().backward()
        print(th, "threads 256^2 iter", i, f"{time.perf_counter()-t0:.2f} s", flush=True)
