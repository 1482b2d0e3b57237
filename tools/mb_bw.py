import torch, time
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
a = torch.empty(1 << 28, device=dev, dtype=torch.float32); b = torch.empty_like(a)
t = timeit(lambda: a.zero_()); print(f"fill  1 GiB: {t*1e3:7.1f} us  {a.numel()*4/t/1e6:6.0f} GB/s write")
t = timeit(lambda: b.copy_(a)); print(f"copy  1 GiB: {t*1e3:7.1f} us  {2*a.numel()*4/t/1e6:6.0f} GB/s read+write")
t = timeit(lambda: a.sum()); print(f"sum   1 GiB: {t*1e3:7.1f} us  {a.numel()*4/t/1e6:6.0f} GB/s read")
h = torch.empty(1 << 29, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: h.zero_()); print(f"fill bf16 1 GiB: {t*1e3:7.1f} us  {h.numel()*2/t/1e6:6.0f} GB/s write")
