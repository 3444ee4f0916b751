"""Diagnostic: achievable HBM write / copy bandwidth on this GPU (torch fill_ / copy_)."""
import torch
x = torch.empty(1 << 28, dtype=torch.float32, device='cuda')   # 1 GiB
y = torch.empty_like(x)
for name, fn, nbytes in (("fill 1 GiB", lambda: x.fill_(1.0), x.numel() * 4), ("copy 1 GiB", lambda: y.copy_(x), 2 * x.numel() * 4)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms*1e3:.0f} us, {nbytes/ms/1e6:.0f} GB/s")
