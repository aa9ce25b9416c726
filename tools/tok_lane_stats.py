"""Lane utilisation of the two decoding passes of k_huffman_decode_tok on the CPU execution model (tests/hipemu built with -DGJ_TOK_STATS):
symbols decoded by the lanes of a wave against 64 x the longest lane (what the wave pays), and how many lanes a storing chunk fills.
usage: tools/tok_lane_stats.py [4k|8k]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
import bench  # noqa: E402
import oracle as O  # noqa: E402
from gpujpeg_amd import libgpujpeg as G  # noqa: E402

w, h = (7680, 4320) if "8k" in sys.argv[1:] else (3840, 2160)
frame = bench.synth_frame(None, w, h, "natural", 12345, torch.device("cpu")).numpy().reshape(-1)
jpeg = O.encode(O.make_image(w, h), frame)
os.environ["GJ_DEC_TOKENS"] = "1"
out = os.path.join(ROOT, "tests", "hipemu", "_build_stats48")
subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tests", "hipemu"), f"OUT={out}", "EXTRA=-DGJ_TOK_STATS -DGJ_TOK_SYNC=48"])
lib = G.Library(os.path.join(out, "libgpujpeg_emu.so"))
assert lib.L.gpujpeg_init_device(0, 0) == 0
dec = G.Decoder(lib)
dec.decode(jpeg)
st = (C.c_ulonglong * 64).in_dll(lib.L, "gj_tok_stats")
print(f"{w}x{h}: {st[0]} sub-sequences")
print("  groups by sub-sequences (buckets of 32): " + " ".join(f"{32 * i}:{st[32 + i]}" for i in range(32) if st[32 + i]))
if st[24]:
    print(f"  storing pass: {st[24]} chunks of {st[25] / st[24]:.1f} sub-sequences for {st[27]} wave quarters that need {st[26]} chunks of 64; "
          f"pool left for the stages: see DESIGN 4.3")
for name, i in (("first pass (without the run-in)", 16), ("storing pass", 20)):
    s, m, n = st[i], st[i + 1], st[i + 2]
    print(f"  {name}: {n} wave passes, {s} symbols, {s / max(1, n):.0f} per wave pass, lanes busy {100.0 * s / max(1, m):.1f}% of 64 x longest lane")
