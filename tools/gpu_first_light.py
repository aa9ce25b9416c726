"""Stage-by-stage parity probe for a GPU box: prints where the HIP path first differs from the oracle."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
from gpujpeg_amd import libgpujpeg as G  # noqa: E402

lib = G.Library()
assert lib.L.gpujpeg_init_device(0, 1) == 0


def natural(w, h, c=3, seed=1):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    chans = [128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0), xx * 255.0 / w, yy * 255.0 / h, 128 + 0 * xx][:c]
    img = np.stack(chans, -1) + rng.normal(0, 6, (h, w, c))
    return np.clip(img, 0, 255).astype(np.uint8).reshape(-1)


def first_diff(a, b):
    n = min(a.size, b.size)
    d = np.nonzero(a[:n] != b[:n])[0]
    return (int(d[0]), int(d.size)) if d.size else (None, 0)


def run(name, w, h, raw, pixfmt=G.P012_444, cs=G.RGB, quality=75, restart=-1, interleaved=0, subsampling=None, opf=None, ocs=None, fused=True):
    print(f"=== {name}: {w}x{h} pf={pixfmt} q={quality} r={restart} il={interleaved} fused={fused}", flush=True)
    try:
        p = lib.default_parameters()
        p.quality, p.restart_interval, p.interleaved, p.verbose = quality, restart, interleaved, -1
        if subsampling is not None:
            lib.L.gpujpeg_parameters_chroma_subsampling(G.C.byref(p), subsampling)
        pi = lib.default_image_parameters()
        pi.width, pi.height, pi.pixel_format, pi.color_space = w, h, pixfmt, cs
        ss = None
        if subsampling is not None:
            ss = [((subsampling >> (28 - 8 * i)) & 15, (subsampling >> (24 - 8 * i)) & 15) for i in range(4)]
            ss = [s for s in ss if s[0] * s[1]]
        I = O.make_image(w, h, pixel_format=pixfmt, color_space=cs, quality=quality, restart_interval=restart, interleaved=interleaved, subsampling=ss)
        planes = O.preprocess(I, raw)
        coefs = O.fdct_quant(I, planes)
        want = O.encode_from_coefs(I, coefs)
        enc = G.Encoder(lib)
        enc.set_fused(fused)
        t0 = time.time()
        jpeg = enc.encode(p, pi, raw)
        t1 = time.time()
        got_coefs = enc.coefficients(I.data_size)
        fd, nd = first_diff(got_coefs, coefs)
        print(f"  encode {1e3*(t1-t0):.2f} ms; coef mismatches: {nd} of {coefs.size}" + (f" first at {fd} (block {fd//64}, pos {fd%64}) got {got_coefs[fd]} want {coefs[fd]}" if nd else ""))
        if nd and not fused:
            gp = enc.planes(I.data_size)
            fdp, ndp = first_diff(gp, planes)
            print(f"  plane mismatches: {ndp}" + (f" first at {fdp} got {gp[fdp]} want {planes[fdp]}" if ndp else ""))
        fd, nd = first_diff(jpeg, want)
        print(f"  jpeg size {jpeg.size} (oracle {want.size}); byte mismatches: {nd}" + (f" first at {fd}: got {jpeg[fd:fd+8]} want {want[fd:fd+8]}" if nd else ""), "EQUAL" if np.array_equal(jpeg, want) else "DIFFERENT")
        # decode the ORACLE stream so that decoder checks do not depend on the encoder
        dec = G.Decoder(lib)
        dec.set_fused(fused)
        dec.keep_coefficients()
        if opf is not None:
            dec.set_output_format(ocs if ocs is not None else G.CS_DEFAULT, opf)
        t0 = time.time()
        px, info = dec.decode(want)
        t1 = time.time()
        wpx, winfo = O.decode(want, -1 if opf is None else opf, -1 if ocs is None else ocs)
        s = O.parse(want, -1 if opf is None else opf, -1 if ocs is None else ocs)
        wc = O.huffman_decode(s, want)
        gc = dec.coefficients(s.img.data_size)
        fd, nd = first_diff(gc, wc)
        print(f"  decode {1e3*(t1-t0):.2f} ms; coef mismatches: {nd}" + (f" first at {fd} (block {fd//64} pos {fd%64}) got {gc[fd]} want {wc[fd]}" if nd else ""))
        fd, nd = first_diff(px, wpx)
        print(f"  pixels {px.size} (oracle {wpx.size}) mismatches: {nd}" + (f" first at {fd} got {px[fd]} want {wpx[fd]}" if nd else ""), "EQUAL" if np.array_equal(px, wpx) else "DIFFERENT")
        enc.close()
        dec.close()
    except Exception:
        traceback.print_exc()


for fused in (False, True):
    run("tiny noise", 64, 64, O.noise(64 * 64 * 3), restart=4, fused=fused)
    run("natural", 640, 368, natural(640, 368), fused=fused)
    run("odd noise", 1119, 561, O.noise(1119 * 561 * 3), fused=fused)
    run("hd gradient r24", 1920, 1080, O.gradient(1920, 1080, 3), restart=24, fused=fused)
run("interleaved 444", 640, 368, natural(640, 368), interleaved=1)
run("uyvy 422 interleaved q90", 642, 366, O.noise(O.raw_size(642, 366, O.PF_422_P1020)), pixfmt=G.P1020_422, cs=G.YCBCR_JPEG, quality=90, interleaved=1, opf=G.P1020_422, ocs=G.YCBCR_JPEG)
run("420 planar from rgb", 322, 242, natural(322, 242), subsampling=G.SUBSAMPLING_420, interleaved=1)
run("420 planar in", 322, 242, O.noise(O.raw_size(322, 242, O.PF_420_P0P1P2)), pixfmt=G.P0P1P2_420, cs=G.YCBCR_JPEG, opf=G.P0P1P2_420, ocs=G.YCBCR_JPEG)
run("gray", 333, 111, O.noise(333 * 111), pixfmt=G.U8, cs=G.YCBCR_JPEG)
run("rgba", 200, 100, O.noise(200 * 100 * 4), pixfmt=G.P0123_4444, cs=G.RGB, subsampling=G.SUBSAMPLING_4444, interleaved=1)
run("restart 0", 200, 120, natural(200, 120), restart=0)
run("restart 1", 64, 48, O.noise(64 * 48 * 3), restart=1)
run("big restart", 1024, 512, natural(1024, 512), restart=300)
run("q100 noise", 256, 256, O.noise(256 * 256 * 3), quality=100, restart=8)
print("done")
