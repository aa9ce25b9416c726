"""Driver of tests/test_sanitizers.py (runs in a subprocess with the sanitizer runtime preloaded): the API tests with hostile input and a
corpus of truncated / damaged image files through the product's file readers, on the sanitizer build of the CPU execution model."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "oracle")]
import oracle as O  # noqa: E402
import test_gpu_api as A  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from conftest import CASES, natural_image  # noqa: E402
from gpujpeg_amd import libgpujpeg as G  # noqa: E402

O.build()
O.lib()
lib = G.Library(sys.argv[1])
assert lib.L.gpujpeg_init_device(0, 0) == 0
mp = pytest.MonkeyPatch()

# ---- parity bodies on the instrumented build (every kernel family once)
for case in CASES[:8] + CASES[12:16]:
    T.test_encode_decode_bit_exact(O, G, lib, case, True)
for tc in T.TOKEN_CASES[:4]:
    T.test_token_mode_decoder(O, G, lib, tc, mp)
for tc in T.TOKEN_422_CASES[:3]:
    T.test_token_mode_decoder_422(O, G, lib, tc, mp)
print("parity bodies ok", flush=True)

# ---- hostile streams
A.test_error_paths(G, lib)
A.test_hostile_tables_and_index(O, G, lib)
A.test_damaged_streams_do_not_crash(O, G, lib)
T.test_token_mode_damaged_streams(O, G, lib, mp)
print("hostile streams ok", flush=True)

# ---- image files: every prefix class of valid TGA / BMP / PNM / PAM / PNG / Y4M files, and files with damaged headers
load = lib.L.gpujpeg_image_load_from_file
load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
destroy = lib.L.gpujpeg_image_destroy
destroy.argtypes = [C.POINTER(C.c_uint8)]
save = lib.L.gpujpeg_image_save_to_file
save.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]
w, h = 37, 21
raw = natural_image(w, h, 3, seed=2)
pi = lib.default_image_parameters()
pi.width, pi.height, pi.pixel_format, pi.color_space = w, h, 1, 1
rng = np.random.default_rng(5)
with tempfile.TemporaryDirectory() as tmp:
    n_files = n_loaded = 0
    for ext in ("tga", "bmp", "pnm", "pam", "png", "y4m"):
        good = os.path.join(tmp, "good." + ext)
        if save(good.encode(), raw.ctypes.data, raw.size, C.byref(pi)) != 0:
            continue
        data = np.fromfile(good, np.uint8)
        variants = [data[:k] for k in sorted({0, 1, 2, 5, 11, 17, 18, 19, 33, 54, 55, 100, data.size // 2, data.size - 1})]
        for t in range(24):  # damaged header bytes
            v = data.copy()
            idx = rng.integers(0, min(64, v.size), size=1 + t % 4)
            v[idx] = rng.integers(0, 256, size=idx.size, dtype=np.uint8)
            variants.append(v)
        for i, v in enumerate(variants):
            p = os.path.join(tmp, f"bad{i}." + ext)
            np.asarray(v, np.uint8).tofile(p)
            img, size = C.POINTER(C.c_uint8)(), C.c_size_t(0)
            n_files += 1
            if load(p.encode(), C.byref(img), C.byref(size)) == 0:
                n_loaded += 1
                destroy(img)
    print(f"image files: {n_files} damaged files, {n_loaded} still loadable", flush=True)
mp.undo()
print("DONE", flush=True)
