"""experiment: decode time of an 8K frame coded without restart markers (one segment per scan)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpujpeg_amd import libgpujpeg as G
from bench import synth_frame
lib = G.Library(); assert lib.L.gpujpeg_init_device(0, 0) == 0
for (w, h) in [(1920, 1080), (7680, 4320)]:
    frame = synth_frame(w, h, "natural", 1, torch.device("cuda", 0)).cpu().numpy().reshape(-1)
    for ri in (0, -1):
        p = lib.default_parameters(); p.restart_interval, p.verbose = ri, -1
        pi = lib.default_image_parameters(); pi.width, pi.height = w, h
        enc, dec = G.Encoder(lib), G.Decoder(lib)
        jpeg = enc.encode(p, pi, frame)
        dec.decode(jpeg)
        t0 = time.perf_counter(); n = 3
        for _ in range(n): px, _i = dec.decode(jpeg)
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n): enc.encode(p, pi, frame)
        et = (time.perf_counter() - t0) / n
        print(f"{w}x{h} restart={ri}: jpeg {jpeg.size} B, encode {et*1e3:.2f} ms, decode {dt*1e3:.2f} ms (host buffers, incl. PCIe)")
