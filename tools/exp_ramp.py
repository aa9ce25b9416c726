"""Find the one-off stall after a cold start: per-frame wall times per pipeline, outliers printed."""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from gpujpeg_amd import libgpujpeg as G
lib = G.Library(); assert lib.L.gpujpeg_init_device(0, 0) == 0
dev = torch.device("cuda", 0)
sp = bench.Spec(lib, sys.argv[1] if len(sys.argv) > 1 else "8k", "natural", 75, dev, 12345)
L = bench.Lanes(lib, sp, dev, 4)
L.warm(2)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
log = [[] for _ in L.lanes]
def worker(i):
    torch.cuda.set_device(0)
    ln = L.lanes[i]
    for k in range(N):
        a = time.perf_counter(); jp, js = L.encode(ln); b = time.perf_counter(); L.decode(ln, jp, js); c = time.perf_counter()
        log[i].append((a, b - a, c - b))
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
[t.start() for t in th]; [t.join() for t in th]
el = time.perf_counter() - t0
print("rate", sp.pixels * 4 * N / el / 1e9)
for i in range(4):
    for k, (a, e, d) in enumerate(log[i]):
        if e > 0.003 or d > 0.003:
            print(f"lane {i} frame {k} at {a - t0:.3f}s enc {e*1e3:.1f} ms dec {d*1e3:.1f} ms")
