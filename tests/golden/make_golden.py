#!/usr/bin/env python3
"""Generates tests/golden/golden.json + a few complete JPEG files from the REFERENCE's own code on the CPU
(oracle/_ref/libgpujpeg_ref.so: /root/reference/src/*.c and the CUDA translation units gpujpeg_preprocessor.cu,
gpujpeg_dct_gpu.cu, gpujpeg_postprocessor.cu compiled where they lie, contraction off; see oracle/Makefile).
Run in the authoring container (needs /root/reference); the outputs are committed so that the restatement (in its
gjo_set_fma(0) mode) can be checked where the reference is absent (GPU box).
The digests of the FUSED arithmetic, which the product reproduces, are produced on the GPU box by the reference's
kernels compiled with hipcc: tests/golden/golden_hip.json, written by tests/test_gpu_refhip.py::test_golden_hip."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
from conftest import CASES, api_params, make_raw  # noqa: E402
from gpujpeg_amd import libgpujpeg as G  # noqa: E402


def main():
    O.build()
    assert O.have_ref(), "needs oracle/_ref (i.e. /root/reference)"
    ref = G.Library(O.REF_PATH)
    golden = {"_comment": "sha256 of the JPEG and of the decoded default-format samples produced by oracle/_ref/libgpujpeg_ref.so (reference host C + reference .cu on the CPU, contraction off) per case of tests/conftest.py:CASES",
              "cases": {}}
    for case in CASES:
        raw = make_raw(O, case)
        p, pi = api_params(ref, G, case)
        jpeg = G.Encoder(ref).encode(p, pi, raw)
        px, info = G.Decoder(ref).decode(jpeg)
        golden["cases"][case[0]] = {
            "raw_sha256": hashlib.sha256(raw.tobytes()).hexdigest(),
            "jpeg_size": int(jpeg.size), "jpeg_sha256": hashlib.sha256(jpeg.tobytes()).hexdigest(),
            "pixels_sha256": hashlib.sha256(px.tobytes()).hexdigest(),
            "out": [info.width, info.height, info.pixel_format, info.color_space],
        }
        if case[0] in ("rgb_1x1", "rgb_7x9", "rgb_q50_r1"):
            with open(os.path.join(HERE, case[0] + ".jpg"), "wb") as f:
                f.write(jpeg.tobytes())
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print(f"wrote {len(golden['cases'])} cases")


if __name__ == "__main__":
    main()
