#!/usr/bin/env python3
"""What does a plain device-to-device copy reach on this box?  torch's copy kernel over buffers of the job's sizes (reads +
writes counted), HIP events, after a spin-up."""
import torch

torch.cuda.set_device(0)
for mb in (128, 512, 1024):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
    b = torch.empty_like(a)
    for _ in range(200):
        b.copy_(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    reps = 100
    for _ in range(reps):
        b.copy_(a)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"copy of {mb} MiB: {ms * 1e3:.1f} us  {2 * n / ms / 1e9:.2f} TB/s (read + write)")
    s.record()
    for _ in range(reps):
        b.fill_(3)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"fill of {mb} MiB: {ms * 1e3:.1f} us  {n / ms / 1e9:.2f} TB/s (write)")
