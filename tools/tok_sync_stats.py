"""How well does the run-in of k_huffman_decode_tok synchronise? Runs the kernel on the CPU execution model (tests/hipemu, built with
-DGJ_TOK_STATS and the given -DGJ_TOK_SYNC) on a synthetic frame of bench.py's generator and prints, per round, the share of
sub-sequences that had to be decoded again. usage: tools/tok_sync_stats.py [sync bits ...]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import oracle as O  # noqa: E402
from gpujpeg_amd import libgpujpeg as G  # noqa: E402

w, h = 3840, 2160
frame = bench.synth_frame(None, w, h, "natural", 12345, torch.device("cpu")).numpy().reshape(-1)
jpeg = O.encode(O.make_image(w, h), frame)
os.environ["GJ_DEC_TOKENS"] = "1"
for sync in [int(a) for a in sys.argv[1:]] or [32, 48, 64, 96, 128]:
    out = os.path.join(ROOT, "tests", "hipemu", f"_build_stats{sync}")
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tests", "hipemu"), f"OUT={out}", f"EXTRA=-DGJ_TOK_STATS -DGJ_TOK_SYNC={sync}"])
    lib = G.Library(os.path.join(out, "libgpujpeg_emu.so"))
    assert lib.L.gpujpeg_init_device(0, 0) == 0
    dec = G.Decoder(lib)
    px, _ = dec.decode(jpeg)
    st = (C.c_ulonglong * 16).in_dll(lib.L, "gj_tok_stats")
    n = st[0]
    print(f"run-in {sync:3d} bits: {n} sub-sequences; decoded again in round 1.. : " + " ".join(f"{100.0 * st[1 + r] / n:.2f}%" for r in range(8) if st[1 + r]) +
          f"; rounds per group (sum {st[15]})", flush=True)
    dec.close()
