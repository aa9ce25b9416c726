"""Where the time of k_huffman_decode_par goes, per workgroup (development aid): apply tools/dbg_decoder_phases.patch (wall-clock stamps at the
phase boundaries, written into the coefficient planes, which token mode leaves unused), rebuild, run this on the GPU box, revert the patch.
Prints start / phase / total times over the workgroups of one 8K decode and their relation to the position in the stream."""
import sys, os, numpy as np, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import torch
from conftest import natural_image
from gpujpeg_amd import libgpujpeg as G
lib = G.Library(); assert lib.L.gpujpeg_init_device(0, 0) == 0
w, h = 7680, 4320
sys.path.insert(0, '.')
import bench
raw = bench.synth_frame(lib, w, h, "natural", 12345, torch.device("cuda:0")).cpu().numpy().reshape(-1)
p = lib.default_parameters(); p.quality = 75; p.restart_interval = G.RESTART_AUTO; p.verbose = -1
pi = lib.default_image_parameters(); pi.width, pi.height = w, h
jpeg = G.Encoder(lib).encode(p, pi, raw)
dec = G.Decoder(lib)
for _ in range(3): dec.decode(jpeg)
n = 1400 * 8
t = dec.coefficients(w * h * 3)[:n * 4].view(np.uint64).reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print("workgroups", len(t), "unit 10 ns")
names = ["start", "setup done", "unstuffed", "seg0 unstuffed", "rounds done", "stored", "dc done", "loads issued"]
for i in range(8):
    x = (t[:, i] - t0) / 100.0
    print(f"{names[i]:14s} abs us: min {x.min():7.1f} mean {x.mean():7.1f} p50 {np.median(x):7.1f} p90 {np.percentile(x,90):7.1f} max {x.max():7.1f}")
for i in range(1, 7):
    d = (t[:, i] - t[:, i-1]) / 100.0
    print(f"phase -> {names[i]:14s} us: mean {d.mean():6.1f} p50 {np.median(d):6.1f} p90 {np.percentile(d,90):6.1f} max {d.max():6.1f}")
tot = (t[:, 6] - t[:, 0]) / 100.0
print("workgroup total us: mean %.1f p50 %.1f p90 %.1f max %.1f; kernel span %.1f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), (t[:, 6].max() - t0) / 100.0))
# per-batch bytes from the stream
j = jpeg
m = np.nonzero((j[:-1] == 0xFF) & (j[1:] >= 0xD0) & (j[1:] <= 0xD7))[0]
sos = np.nonzero((j[:-1] == 0xFF) & (j[1:] == 0xDA))[0]
print("rst markers", m.size, "sos", sos.size)
# segment boundaries in stream order: scan starts + rst markers + scan ends (approx: use marker positions only)
bounds = np.sort(np.concatenate([m, sos + 14, [j.size - 2]]))
seglen = np.diff(bounds)
G = 32  # (only for the uniform plan)
nb = len(t)
bb = np.array([seglen[i * G:(i + 1) * G].sum() for i in range(nb)])
mx = np.array([seglen[i * G:(i + 1) * G].max() if len(seglen[i * G:(i + 1) * G]) else 0 for i in range(nb)])
un = (t[:, 2] - t[:, 1]) / 100.0
st = (t[:, 0] - t0) / 100.0
order = np.argsort(bb)
for q in range(0, nb, nb // 10):
    sel = order[q:q + nb // 10]
    print(f"batch bytes {bb[sel].mean():7.0f} max seg {mx[sel].mean():6.0f}  unstuff us {un[sel].mean():6.1f}  rounds {((t[sel,4]-t[sel,2])/100).mean():5.1f}  store {((t[sel,5]-t[sel,4])/100).mean():5.1f}  start {st[sel].mean():6.1f}  total {((t[sel,6]-t[sel,0])/100).mean():6.1f}")
print("by workgroup index:")
for q in range(0, nb, nb // 10):
    sel = np.arange(q, min(nb, q + nb // 10))
    print(f"wg {q:5d}.. bytes {bb[sel].mean():7.0f} unstuff us {un[sel].mean():6.1f} start {st[sel].mean():6.1f} total {((t[sel,6]-t[sel,0])/100).mean():6.1f}")
