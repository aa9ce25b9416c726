"""Find the one-off stall after a cold start: per-frame wall times per pipeline, outliers printed.
usage: exp_ramp.py [workload] [frames] [perf_stats 0|1] [streams]   (runtime switches are taken from the environment: the caller's A/B)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from gpujpeg_amd import libgpujpeg as G
lib = G.Library(); assert lib.L.gpujpeg_init_device(0, 0) == 0
dev = torch.device("cuda", 0)
sp = bench.Spec(lib, sys.argv[1] if len(sys.argv) > 1 else "8k", "natural", 75, dev, 12345)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
stats = int(sys.argv[3]) if len(sys.argv) > 3 else 1
S = int(sys.argv[4]) if len(sys.argv) > 4 else 4
L = bench.Lanes(lib, sp, dev, S)
L.warm(2)
if not stats:
    L.set_stats(False)
import gc
GCLOG = []
if os.environ.get("EXP_GC") == "off":
    gc.disable()
elif os.environ.get("EXP_GC") == "freeze":
    gc.collect(); gc.freeze()
_t = [0.0]
def _cb(phase, info):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        GCLOG.append((time.perf_counter(), info["generation"], time.perf_counter() - _t[0], info["collected"]))
gc.callbacks.append(_cb)
log = [[] for _ in L.lanes]
def worker(i):
    torch.cuda.set_device(0)
    ln = L.lanes[i]
    for k in range(N):
        a = time.perf_counter(); jp, js = L.encode(ln); b = time.perf_counter(); L.decode(ln, jp, js); c = time.perf_counter()
        log[i].append((a, b - a, c - b))
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(i,)) for i in range(S)]
[t.start() for t in th]; [t.join() for t in th]
el = time.perf_counter() - t0
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith(("GPU_MAX", "ROC_", "HSA_KERNARG", "HIP_FORCE", "DEBUG_", "HSA_NO", "HSA_ENABLE_SCR", "AMD_DIRECT")))
print(f"[{tag or 'defaults'}] stats={stats} streams={S} frames={N} rate {sp.pixels * S * N / el / 1e9:.1f} Gpix/s")
print("   gc mode", os.environ.get("EXP_GC", "default"), "counts", gc.get_count(), "objects", len(gc.get_objects()))
for (t, gen, dur, n) in GCLOG:
    if dur > 0.002:
        print(f"   gc generation {gen} at {t - t0:.3f}s took {dur*1e3:.1f} ms, collected {n}")
for i in range(S):
    for k, (a, e, d) in enumerate(log[i]):
        if e > 0.005 or d > 0.005:
            print(f"   lane {i} frame {k} at {a - t0:.3f}s enc {e*1e3:.1f} ms dec {d*1e3:.1f} ms")
