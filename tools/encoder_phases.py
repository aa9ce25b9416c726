"""Where the time of k_encode_rgb444 goes, per workgroup (development aid; VERDICT r2 #5 asked what keeps the kernel from 68 us). Needs the trace
build of the library (`make -C gpujpeg_amd/csrc trace` -> gpujpeg_amd/lib/libgpujpeg_trace.so); run on the GPU box. Prints start / phase / total
times over the workgroups of one encode of bench.py's 8K frame (or --workload)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from gpujpeg_amd import libgpujpeg as G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="8k")
args = ap.parse_args()
lib = G.Library(os.environ.get("GJ_TRACE_LIB") or os.path.join(ROOT, "gpujpeg_amd", "lib", "libgpujpeg_trace.so"))
assert lib.L.gpujpeg_init_device(0, 0) == 0
dev = torch.device("cuda", 0)
spec = bench.Spec(lib, args.workload, "natural", 75, dev, 12345)
L = bench.Lanes(lib, spec, dev, 1)
ln = L.lanes[0]
for _ in range(3):
    L.encode(ln)
torch.cuda.synchronize()
NWG, SLOTS = 8192, 16
buf = torch.zeros(NWG * SLOTS, dtype=torch.int64, device=dev)
lib.L.gj_hip_trace_set_encoder.argtypes = [C.c_void_p]
assert lib.L.gj_hip_trace_set_encoder(buf.data_ptr()) == 0
L.encode(ln)
torch.cuda.synchronize()
assert lib.L.gj_hip_trace_set_encoder(None) == 0
t = buf.cpu().numpy().reshape(NWG, SLOTS)
t = t[t[:, 0] > 0]
names = ["start", "pixels loaded + converted"]
for c in ("Y", "Cb", "Cr"):
    names += [f"{c}: transformed", f"{c}: walk done", f"{c}: positions", f"{c}: merged + drained"]
t0 = t[:, 0].min()
n = len(names)
print(f"{args.workload}: workgroups {len(t)}; times in us (100 MHz clock), stamps of the first work-item of every workgroup")
x = (t[:, 0] - t0) / 100.0
print(f"  start                         : min {x.min():6.1f} p10 {np.percentile(x, 10):6.1f} p50 {np.median(x):6.1f} p90 {np.percentile(x, 90):6.1f} max {x.max():6.1f}")
for i in range(1, n):
    d = (t[:, i] - t[:, i - 1]) / 100.0
    print(f"  -> {names[i]:27s}: mean {d.mean():6.1f} p50 {np.median(d):6.1f} p90 {np.percentile(d, 90):6.1f} max {d.max():6.1f}")
tot = (t[:, n - 1] - t[:, 0]) / 100.0
end = (t[:, n - 1] - t0) / 100.0
print("  workgroup total: mean %.1f p50 %.1f p90 %.1f max %.1f; kernel span %.1f; workgroups that start after 40 us: %d, their mean total %.1f" %
      (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), end.max(), int((x > 40).sum()), float(tot[x > 40].mean()) if (x > 40).any() else 0.0))
busy = np.zeros(int(end.max()) + 2)
for a, b in zip(x.astype(int), end.astype(int)):
    busy[a:b + 1] += 1
print("  resident workgroups over time (every 10 us): " + " ".join(f"{int(busy[i])}" for i in range(0, len(busy), 10)))
