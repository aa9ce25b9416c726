"""Bench fixture `camera_bt709_422_q95.jpg`: the reference's camera sample (colors/camera_bt709_422.yuv: one 1920x1080 frame, packed
UYVY 4:2:2, BT.709 limited range) as a q95 4:2:2 interleaved JPEG coded by the CPU oracle. /root/reference does not exist on the GPU
box, so the frame travels in this form; bench.py decodes it ONCE with the product's decoder and tiles the RGB result to the size of
the workload (`--pattern camera`, workload entry `8k_camera`).

    python tests/golden/make_camera_fixture.py            (in the build container, where /root/reference is present)"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import oracle as O  # noqa: E402

SRC = "/root/reference/colors/camera_bt709_422.yuv"
W, H, QUALITY = 1920, 1080, 95
P1020_422, YCBCR_BT709 = 3, 4

if __name__ == "__main__":
    raw = np.fromfile(SRC, np.uint8)
    assert raw.size == W * H * 2
    img = O.make_image(W, H, pixel_format=P1020_422, color_space=YCBCR_BT709, quality=QUALITY, interleaved=1)
    jpeg = O.encode(img, raw)
    out = os.path.join(HERE, "camera_bt709_422_q95.jpg")
    jpeg.tofile(out)
    print(out, jpeg.size, "bytes, sha256", hashlib.sha256(jpeg.tobytes()).hexdigest())
