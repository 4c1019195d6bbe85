#!/usr/bin/env python3
"""Streaming-bandwidth calibration on the GPU box (torch kernels, HIP events): what fraction of the
nominal 8 TB/s a plain read / copy reaches on this part, to put the roofline fractions in context."""
import torch

def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

for mb in (664, 4096):
    n = mb * 1024 * 1024 // 8
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    t = timed(lambda: x.sum())
    print(f"read  {mb:5d} MB: {mb*1.048576e6/t/1e12:.2f} TB/s ({t*1e6:.0f} us)")
    t = timed(lambda: y.copy_(x))
    print(f"copy  {mb:5d} MB: {2*mb*1.048576e6/t/1e12:.2f} TB/s read+write ({t*1e6:.0f} us)")
    t = timed(lambda: y.fill_(1.0))
    print(f"write {mb:5d} MB: {mb*1.048576e6/t/1e12:.2f} TB/s ({t*1e6:.0f} us)")
