#!/usr/bin/env python3
"""PCIe rates of the box (pinned host memory): D2H alone, H2D alone, both at once on two streams, by chunk size.
The floor of the end-to-end span is the D2H row."""
import time
import torch

dev = torch.device("cuda:0")
total = 768 << 20
src = torch.empty(total, dtype=torch.uint8, device=dev).fill_(1)
dst = torch.empty(total, dtype=torch.uint8, device=dev)
hd = torch.empty(total, dtype=torch.uint8).pin_memory()
hu = torch.empty(total, dtype=torch.uint8).pin_memory().fill_(2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(d2h, h2d, chunk):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for o in range(0, total, chunk):
        if d2h:
            with torch.cuda.stream(s1):
                hd[o:o + chunk].copy_(src[o:o + chunk], non_blocking=True)
        if h2d:
            with torch.cuda.stream(s2):
                dst[o:o + chunk].copy_(hu[o:o + chunk], non_blocking=True)
    torch.cuda.synchronize()
    return total / (time.perf_counter() - t) / 1e9


for chunk in (4 << 20, 32 << 20, 768 << 20):
    for _ in range(2):
        a, b, c = run(True, False, chunk), run(False, True, chunk), run(True, True, chunk)
    print(f"chunk {chunk >> 20:4d} MB: D2H {a:5.1f} GB/s   H2D {b:5.1f} GB/s   both at once {c:5.1f} GB/s each way")
