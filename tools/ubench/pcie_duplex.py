#!/usr/bin/env python3
"""Does the host link of this box carry both directions at once? (the host-staged batch's target assumes 57 GB/s EACH way.)
Pinned host buffers, one HIP stream per direction, hipMemcpyAsync of 256 MiB pieces: each direction alone, then both at the same time."""
import time
import torch

N = 256 << 20
h_in = torch.empty(N, dtype=torch.uint8).pin_memory()
h_out = torch.empty(N, dtype=torch.uint8).pin_memory()
d_a = torch.empty(N, dtype=torch.uint8, device="cuda")
d_b = torch.ones(N, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(h2d, d2h, reps=12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1):
                d_a.copy_(h_in, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2):
                h_out.copy_(d_b, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return reps * N / dt / 1e9


for _ in range(2):
    run(True, True, 2)
a = run(True, False)
b = run(False, True)
c = run(True, True)
print(f"host -> device alone {a:6.1f} GB/s   device -> host alone {b:6.1f} GB/s   both at once {c:6.1f} GB/s each way = {2 * c:6.1f} GB/s in total")
# the same with the copies cut into pieces (24.9 MB = one 4K RGB frame), k streams per direction
def pieces(P, k, reps=6, h2d=True, d2h=True):
    ss = [torch.cuda.Stream() for _ in range(2 * k)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for r in range(reps):
        for i in range(N // P):
            if h2d:
                with torch.cuda.stream(ss[i % k]):
                    d_a[i * P:(i + 1) * P].copy_(h_in[i * P:(i + 1) * P], non_blocking=True)
            if d2h:
                with torch.cuda.stream(ss[k + i % k]):
                    h_out[i * P:(i + 1) * P].copy_(d_b[i * P:(i + 1) * P], non_blocking=True)
            n += P
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0) / 1e9


for P in (3840 * 2160 * 3, 1 << 20, 1920 * 1080 * 3):
    for k in (1, 2, 3, 4):
        pieces(P, k, 1)
        print(f"pieces of {P / 1e6:5.1f} MB, {k} stream(s) per direction: both {pieces(P, k):6.1f} GB/s each way; host -> device only {pieces(P, k, 6, True, False):6.1f}, device -> host only {pieces(P, k, 6, False, True):6.1f}")
