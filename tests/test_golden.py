"""The committed golden vectors against the CPU oracle; runs without /root/reference and without a GPU.
tests/golden/golden.json: produced by the reference's own code on the CPU with contraction off (make_golden.py) -- the restatement
must reproduce it in its gjo_set_fma(0) mode. tests/golden/golden_hip.json: produced on an MI355X by the reference's own kernels
compiled with hipcc (tests/test_gpu_refhip.py) -- the restatement must reproduce it with its pinned fusion map."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import CASES, make_raw, oracle_image

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))["cases"]
HIP_PATH = os.path.join(HERE, "golden", "golden_hip.json")
GOLDEN_HIP = json.load(open(HIP_PATH))["cases"] if os.path.exists(HIP_PATH) else None


@pytest.fixture
def nofma(O):
    O.lib().gjo_set_fma(0)
    yield
    O.lib().gjo_set_fma(1)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_golden(O, nofma, case):
    _check(O, case, GOLDEN[case[0]])


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_golden_hip(O, case):
    if GOLDEN_HIP is None:
        pytest.skip("tests/golden/golden_hip.json not produced yet (GPU box)")
    _check(O, case, GOLDEN_HIP[case[0]])


def _check(O, case, g):
    raw = make_raw(O, case)
    assert hashlib.sha256(raw.tobytes()).hexdigest() == g["raw_sha256"], "input generator drifted"
    jpeg = O.encode(oracle_image(O, case), raw)
    assert jpeg.size == g["jpeg_size"]
    assert hashlib.sha256(jpeg.tobytes()).hexdigest() == g["jpeg_sha256"]
    px, info = O.decode(jpeg)
    assert [info.width, info.height, info.pixel_format, info.color_space] == g["out"]
    assert hashlib.sha256(px.tobytes()).hexdigest() == g["pixels_sha256"]


@pytest.mark.parametrize("name", ["rgb_1x1", "rgb_7x9", "rgb_q50_r1"])
def test_golden_files_decode(O, nofma, name):
    jpeg = np.fromfile(os.path.join(HERE, "golden", name + ".jpg"), np.uint8)
    px, _ = O.decode(jpeg)
    assert hashlib.sha256(px.tobytes()).hexdigest() == GOLDEN[name]["pixels_sha256"]


def test_independent_decoder_agrees(O):
    """An unrelated JPEG implementation (Pillow / libjpeg-turbo) accepts our streams and lands within
    rounding distance of the oracle's pixels (SURVEY 8c iii)."""
    import io
    from PIL import Image
    from conftest import natural_image, psnr
    w, h = 320, 200
    raw = natural_image(w, h)
    for il in (0, 1):
        jpeg = O.encode(O.make_image(w, h, interleaved=il), raw)
        pil = np.asarray(Image.open(io.BytesIO(jpeg.tobytes())).convert("RGB")).reshape(-1)
        ours, _ = O.decode(jpeg)
        assert psnr(pil, ours) > 45.0
        assert abs(psnr(pil, raw) - psnr(ours, raw)) < 0.1
