#!/usr/bin/env python3
"""What does a pure write stream reach on this MI355X?  (calibrates write-dominated kernels such as the CUBIC enlargement)
torch fill / copy over the upscale benchmark's destination size, HIP-event timed."""
import torch

n = 512 * 1080 * 1920 * 4
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n // 4, dtype=torch.uint8, device="cuda")


def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ai = a.view(torch.int32)
t = timed(lambda: ai.fill_(0x01020304))
print("fill  %.2f GB in %.3f ms = %.2f TB/s written" % (n / 1e9, t, n / t / 1e9))
t = timed(lambda: a.zero_())
print("zero  %.2f GB in %.3f ms = %.2f TB/s written" % (n / 1e9, t, n / t / 1e9))
h = n // 2
t = timed(lambda: a[:h].copy_(a[h:]))
print("copy  %.2f GB -> %.2f GB in %.3f ms = %.2f TB/s read+written" % (h / 1e9, h / 1e9, t, 2 * h / t / 1e9))
