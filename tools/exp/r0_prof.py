import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpujpeg_amd import libgpujpeg as G
from bench import synth_frame
lib = G.Library(); assert lib.L.gpujpeg_init_device(0, 0) == 0
w, h = 1920, 1080
frame = synth_frame(w, h, "natural", 1, torch.device("cuda", 0)).cpu().numpy().reshape(-1)
p = lib.default_parameters(); p.restart_interval, p.verbose = 0, -1
pi = lib.default_image_parameters(); pi.width, pi.height = w, h
enc, dec = G.Encoder(lib), G.Decoder(lib)
jpeg = enc.encode(p, pi, frame)
dec.decode(jpeg); dec.decode(jpeg)
