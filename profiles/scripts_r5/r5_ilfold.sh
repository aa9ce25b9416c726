#!/bin/bash
# interleaved scans on the speculative path without the k_marker_table launch: tests, then HD 4:2:0 / 4:2:2 interleaved frames (one call per frame)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -3 ) | tee gpurun_out/r5_gpu_tests.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r5_ilfold.txt
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
from gpujpeg_amd import libgpujpeg as G
import ctypes as C
lib = G.Library(); assert lib.L.gpujpeg_init_device(0, 0) == 0
w, h = 1920, 1080
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:h, 0:w]
img = np.stack([128 + 90 * np.sin(xx / 37.0) * np.cos(yy / 23.0), xx * 255.0 / w, yy * 255.0 / h], -1) + rng.normal(0, 5, (h, w, 3))
raw = np.clip(img, 0, 255).astype(np.uint8).reshape(-1)
for name, sub in (("4:2:0 interleaved", G.SUBSAMPLING_420), ("4:2:2 interleaved", G.SUBSAMPLING_422)):
    p, pi = lib.default_parameters(), lib.default_image_parameters()
    pi.width, pi.height = w, h
    p.interleaved = 1; p.restart_interval = G.RESTART_AUTO
    lib.L.gpujpeg_parameters_chroma_subsampling(C.byref(p), sub)
    enc = G.Encoder(lib); jpeg = enc.encode(p, pi, raw)
    dj = torch.from_numpy(jpeg.copy()).cuda(); out = torch.empty(w * h * 3, dtype=torch.uint8, device='cuda')
    for env in ("1", ""):
        if env: os.environ["GJ_DEC_NO_SPEC"] = "1"
        else: os.environ.pop("GJ_DEC_NO_SPEC", None)
        dec = G.Decoder(lib)
        o = G.DecoderOutput(); o.type, o.data = G.DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, out.data_ptr()
        def one():
            assert lib.L.gpujpeg_decoder_decode(dec.h, C.c_void_p(dj.data_ptr()), jpeg.size, C.byref(o)) == 0
        for _ in range(50): one()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(400): one()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
        print(f"HD {name}, {jpeg.size} B, one decoder, {'careful path (four launches)' if env else 'speculative path'}: {dt * 1e6:7.1f} us per decode, path counters {dec.path_counters()}")
        dec.close()
    enc.close()
PY
