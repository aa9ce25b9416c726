"""experiment: which kind of stream damage crashes the decoder? one subprocess per case"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) == 1:
    for name in "abc":
        for trial in range(6):
            for spec in ("0", "1"):
                r = subprocess.run([sys.executable, __file__, name, str(trial), spec], capture_output=True, text=True)
                tail = (r.stderr.strip().splitlines() or [""])[-1][:150]
                print(name, trial, "spec" if spec == "1" else "careful", "rc", r.returncode, r.stdout.strip()[-60:], "|", tail if r.returncode else "", flush=True)
    sys.exit(0)
import oracle as O
from gpujpeg_amd import libgpujpeg as G
from conftest import natural_image, oracle_image
name, trial, spec = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1"
lib = G.Library(); assert lib.L.gpujpeg_init_device(0, 0) == 0
cfg = {"a": (320, 240, -1, 0, None), "b": (256, 192, 0, 0, None), "c": (320, 240, 3, 1, [(2, 2), (1, 1), (1, 1)])}[name]
w, h, ri, il, ss = cfg
jpeg = O.encode(oracle_image(O, (name, w, h, 1, 1, 80, ri, il, ss, 3)), natural_image(w, h, 3, seed=w))
rng = np.random.default_rng(5 + trial)
bad = jpeg.copy(); hdr = 700
if trial < 3:
    idx = rng.integers(hdr, bad.size - 2, size=40); bad[idx] = rng.integers(0, 255, size=idx.size, dtype=np.uint8)
elif trial == 3:
    for i in rng.integers(hdr, bad.size - 4, size=5): bad[i], bad[i + 1] = 0xFF, 0xD0 + int(rng.integers(0, 8))
elif trial == 4:
    bad = bad[: bad.size // 2].copy()
else:
    bad[hdr + 50:] = 0
dec = G.Decoder(lib)
if spec:
    dec.decode(jpeg)  # primes the header cache: the damaged stream then takes the speculative path
try:
    dec.decode(bad); print("decoded")
except Exception as e:
    print("error return")
px, _ = dec.decode(jpeg)
print("ok" if np.array_equal(px, O.decode(jpeg)[0]) else "WRONG AFTER")
