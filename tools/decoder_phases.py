"""Where the time of k_huffman_decode_tok goes, per workgroup (development aid). Needs the trace build of the library
(`make -C gpujpeg_amd/csrc trace` -> gpujpeg_amd/lib/libgpujpeg_trace.so: wall-clock stamps at the phase boundaries); run on the GPU box.
Prints start / phase / total times over the workgroups of one decode of bench.py's 8K frame (or --workload)."""
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
ap.add_argument("--pattern", default="natural")
ap.add_argument("--quality", type=int, default=75)
args = ap.parse_args()
lib = G.Library(os.environ.get("GJ_TRACE_LIB") or os.path.join(ROOT, "gpujpeg_amd", "lib", "libgpujpeg_trace.so"))
assert lib.L.gpujpeg_init_device(0, 0) == 0
dev = torch.device("cuda", 0)
spec = bench.Spec(lib, args.workload, args.pattern, args.quality, dev, 12345)
L = bench.Lanes(lib, spec, dev, 1)
ln = L.lanes[0]
jp, js = L.encode(ln)
for _ in range(3):
    L.decode(ln, jp, js)
torch.cuda.synchronize()
NWG, SLOTS = 4096, 16
buf = torch.zeros(NWG * SLOTS, dtype=torch.int64, device=dev)
lib.L.gj_hip_trace_set.argtypes = [C.c_void_p]
assert lib.L.gj_hip_trace_set(buf.data_ptr()) == 0
L.decode(ln, jp, js)
torch.cuda.synchronize()
assert lib.L.gj_hip_trace_set(None) == 0
t = buf.cpu().numpy().reshape(NWG, SLOTS)
t = t[t[:, 0] > 0]
names = ["start", "group chosen", "unstuffed", "sub-sequence table", "first pass", "rounds", "prefix sums", "stored + flushed", "DC / records"]
t0 = t[:, 0].min()
print(f"{args.workload} q{args.quality} {args.pattern}: JPEG {js} B, workgroups {len(t)}; times in us (100 MHz clock)")
for i in range(9):
    x = (t[:, i] - t0) / 100.0
    print(f"  at {names[i]:20s}: min {x.min():7.1f} mean {x.mean():7.1f} p50 {np.median(x):7.1f} p90 {np.percentile(x, 90):7.1f} max {x.max():7.1f}")
for i in range(1, 9):
    d = (t[:, i] - t[:, i - 1]) / 100.0
    print(f"  phase -> {names[i]:20s}: mean {d.mean():6.1f} p50 {np.median(d):6.1f} p90 {np.percentile(d, 90):6.1f} max {d.max():6.1f}")
tot = (t[:, 8] - t[:, 0]) / 100.0
print("  workgroup total: mean %.1f p50 %.1f p90 %.1f max %.1f; kernel span %.1f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), (t[:, 8].max() - t0) / 100.0))
slow = np.argsort(tot)[-max(1, len(t) // 20):]  # the slowest 5 % of the workgroups: what sets them apart?
ph = np.diff(t[:, :9], axis=1) / 100.0
print("  slowest 5 %: workgroups " + ", ".join(f"{int(i)}" for i in sorted(slow)[:12]) + " ...; their phases minus everybody's mean: " +
      " ".join(f"{x:+5.1f}" for x in (ph[slow].mean(0) - ph.mean(0))))
n = len(t)
for q in range(0, n, max(1, n // 8)):
    sel = np.arange(q, min(n, q + max(1, n // 8)))
    d = np.diff(t[sel][:, :9], axis=1).mean(0) / 100.0
    print(f"  wg {q:5d}..: start {((t[sel, 0] - t0) / 100).mean():6.1f} phases " + " ".join(f"{x:5.1f}" for x in d) + f" total {d.sum():6.1f}")
# ---- the same for the marker scan (k_markers)
mb = torch.zeros(16384 * SLOTS, dtype=torch.int64, device=dev)
lib.L.gj_hip_trace_set_markers.argtypes = [C.c_void_p]
assert lib.L.gj_hip_trace_set_markers(mb.data_ptr()) == 0
L.decode(ln, jp, js)
torch.cuda.synchronize()
assert lib.L.gj_hip_trace_set_markers(None) == 0
m = mb.cpu().numpy().reshape(-1, SLOTS)
m = m[m[:, 0] > 0]
m0 = m[:, 0].min()
mn = ["start", "bytes read, markers found", "marker list built", "records read", "scans derived", "entries written", "summary out"]
print(f"k_markers: workgroups {len(m)}")
for i in range(7):
    x = (m[:, i] - m0) / 100.0
    print(f"  at {mn[i]:28s}: min {x.min():6.1f} mean {x.mean():6.1f} p90 {np.percentile(x, 90):6.1f} max {x.max():6.1f}")
L.close()
