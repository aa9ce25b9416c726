#!/usr/bin/env python3
"""Encoder GPU time of the subsampled / planar layouts at 4K: the fused lane-per-block kernel (k_encode_blocks) against the generic chain
k_preprocess / k_copy_planes_in + k_dct + k_huffman (gpujpeg_amd_encoder_set_fused(0)). Frames resident in HBM; times are the hipEvent
durations between the first and the last kernel of a call (gpujpeg_amd_encoder_get_kernel_times), averaged over 20 calls."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpujpeg_amd import libgpujpeg as G  # noqa: E402

W, H = (int(os.environ.get("BF_W", 3840)), int(os.environ.get("BF_H", 2160)))
CASES = [  # name, pixel format, colour space, interleaved, subsampling factors or None
    ("rgb_to_420_interleaved", G.P012_444, G.RGB, 1, (2, 2, 1, 1, 1, 1)),
    ("rgb_to_422_interleaved", G.P012_444, G.RGB, 1, (2, 1, 1, 1, 1, 1)),
    ("rgb_to_420_non_interleaved", G.P012_444, G.RGB, 0, (2, 2, 1, 1, 1, 1)),
    ("planar420_in_non_interleaved", G.P0P1P2_420, G.YCBCR_JPEG, 0, None),
    ("planar420_in_interleaved", G.P0P1P2_420, G.YCBCR_JPEG, 1, None),
    ("planar422_in_interleaved", G.P0P1P2_422, G.YCBCR_JPEG, 1, None),
    ("rgb_444_interleaved", G.P012_444, G.RGB, 1, None),
    ("gray", G.U8, G.YCBCR_JPEG, 0, None),
]


def main():
    lib = G.Library()
    assert lib.L.gpujpeg_init_device(0, 0) == 0
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    yy = torch.arange(H, device=dev, dtype=torch.float32).view(-1, 1)
    xx = torch.arange(W, device=dev, dtype=torch.float32).view(1, -1)
    base = (128 + 70 * torch.sin(xx / 97.0) * torch.cos(yy / 61.0) + 3 * torch.randn((H, W), device=dev, generator=g)).clamp(0, 255).to(torch.uint8)
    out = {}
    for name, pf, cs, il, ss in CASES:
        pi = lib.default_image_parameters()
        pi.width, pi.height, pi.pixel_format, pi.color_space = W, H, pf, cs
        n = lib.image_size(pi)
        raw = base.reshape(-1).repeat(4)[:n].contiguous()
        p = lib.default_parameters()
        p.quality, p.restart_interval, p.interleaved, p.verbose, p.perf_stats = 75, G.RESTART_AUTO, il, -1, 1
        if ss:
            lib.L.gpujpeg_parameters_chroma_subsampling(C.byref(p), G.MK_SUBSAMPLING(*ss))
        res = {}
        sizes = {}
        for fused in (1, 0):
            enc = G.Encoder(lib)
            enc.set_fused(fused)
            assert enc.set_option("enc_opt_out", "enc_out_val_device") == 0
            for _ in range(3):
                jp, js = enc.encode_noclone(p, pi, raw.data_ptr(), gpu=True)
            acc = 0.0
            for _ in range(20):
                jp, js = enc.encode_noclone(p, pi, raw.data_ptr(), gpu=True)
                torch.cuda.synchronize()
                acc += sum(enc.kernel_times())
            res["fused_ms" if fused else "generic_ms"] = round(acc / 20, 4)
            jt = torch.empty(js, dtype=torch.uint8, device=dev)
            C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(jt.data_ptr()), C.cast(jp, C.c_void_p), C.c_size_t(js), 3)
            sizes[fused] = jt.cpu().numpy().tobytes()
            enc.close()
        res["speedup"] = round(res["generic_ms"] / res["fused_ms"], 2)
        res["identical_streams"] = sizes[0] == sizes[1]
        res["jpeg_bytes"] = len(sizes[1])
        out[name] = res
        print(name, res, flush=True)
    print(json.dumps({"tool": "bench_formats", "width": W, "height": H, "quality": 75, "results": out}))


if __name__ == "__main__":
    main()
