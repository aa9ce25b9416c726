"""-m gpu: the HIP path, called through the libgpujpeg C ABI, against the CPU oracle and the committed golden
vectors. Integer/byte work must be bit-exact: JPEG bytes, quantised coefficients, decoded pixels."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import CASES, api_params, make_raw, natural_image, oracle_image, psnr

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
# digests produced on an MI355X by the reference's own kernels compiled with hipcc (tests/test_gpu_refhip.py::test_golden_hip)
_GOLDEN_PATH = os.path.join(HERE, "golden", "golden_hip.json")
GOLDEN = json.load(open(_GOLDEN_PATH))["cases"] if os.path.exists(_GOLDEN_PATH) else None


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "generic"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_encode_decode_bit_exact(O, G, gpu_lib, case, fused):
    raw = make_raw(O, case)
    img = oracle_image(O, case)
    planes = O.preprocess(img, raw)
    coefs = O.fdct_quant(img, planes)
    want = O.encode_from_coefs(img, coefs)
    p, pi = api_params(gpu_lib, G, case)
    enc = G.Encoder(gpu_lib)
    enc.set_fused(fused)
    if fused:  # default product path: pixels -> entropy-coded segments in one kernel where the format allows, no coefficient planes
        assert np.array_equal(enc.encode(p, pi, raw), want), "JPEG bytes differ (fully fused path)"
    enc.keep_coefficients()
    jpeg = enc.encode(p, pi, raw)
    assert np.array_equal(enc.coefficients(img.data_size), coefs), "quantised coefficients differ"
    assert np.array_equal(jpeg, want), "JPEG bytes differ"
    assert GOLDEN is not None, "tests/golden/golden_hip.json is missing"
    g = GOLDEN[case[0]]
    assert hashlib.sha256(jpeg.tobytes()).hexdigest() == g["jpeg_sha256"], "differs from the reference-produced golden stream"
    dec = G.Decoder(gpu_lib)
    dec.set_fused(fused)
    dec.keep_coefficients()
    px, info = dec.decode(want)
    s = O.parse(want)
    assert np.array_equal(dec.coefficients(s.img.data_size), O.huffman_decode(s, want)), "entropy decoder differs"
    O.lib().gjo_stream_free(C.byref(s))
    assert hashlib.sha256(px.tobytes()).hexdigest() == g["pixels_sha256"], "pixels differ from the reference-decoded golden"
    assert [info.width, info.height, info.pixel_format, info.color_space] == g["out"]
    enc.close()
    dec.close()


@pytest.mark.parametrize("case", [c for c in CASES if c[6] != 0][:6], ids=lambda c: c[0])
def test_segment_info(O, G, gpu_lib, case):
    raw = make_raw(O, case)
    want = O.encode(oracle_image(O, case, segment_info=1), raw)
    p, pi = api_params(gpu_lib, G, case, segment_info=1)
    jpeg = G.Encoder(gpu_lib).encode(p, pi, raw)
    assert np.array_equal(jpeg, want)
    px, _ = G.Decoder(gpu_lib).decode(jpeg)  # consumes the APP13 index instead of scanning for RSTn
    assert np.array_equal(px, O.decode(want)[0])


@pytest.mark.parametrize("pf,w,h", [(1, 641, 481), (3, 322, 77), (5, 33, 35), (4, 33, 35), (2, 10, 10), (0, 99, 3), (6, 17, 9)])
def test_output_formats(O, G, gpu_lib, pf, w, h):
    cs = 1 if pf in (1, 6) else 3
    raw = O.noise(O.raw_size(w, h, pf), seed=pf + w)
    case = ("x", w, h, pf, cs, 80, 4, 1 if pf != 0 else 0, None, 3)
    jpeg = O.encode(oracle_image(O, case), raw)
    for opf, ocs in [(pf, cs), (1, 1), (G.PIXFMT_NATIVE, G.NONE), (2, 3), (5, 4)]:
        if jpeg is None:
            continue
        dec = G.Decoder(gpu_lib)
        dec.set_output_format(ocs, opf)
        if pf == 0 and opf in (2, 5):
            continue  # planar colour output of a grayscale stream is rejected by the reference too
        px, info = dec.decode(jpeg)
        opx, _ = O.decode(jpeg, info.pixel_format, info.color_space)
        assert np.array_equal(px, opx), (opf, ocs)


def test_exhaustive_colour_transform(O, G, gpu_lib):
    """All 2^24 RGB triples through the preprocessor (generic path) -> planes equal the oracle's; covers the
    integer colour matrices of src/gpujpeg_colorspace.h for every supported internal colour space."""
    w = h = 4096
    v = np.arange(w * h, dtype=np.uint32)
    raw = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(-1)
    for csi in (3, 2, 4):
        case = ("x", w, h, 1, 1, 90, 36, 0, None, csi)
        img = oracle_image(O, case)
        p, pi = api_params(gpu_lib, G, case)
        enc = G.Encoder(gpu_lib)
        enc.set_fused(False)
        enc.encode(p, pi, raw)
        assert np.array_equal(enc.planes(img.data_size), O.preprocess(img, raw)), csi
        enc.set_fused(True)
        enc.keep_coefficients()
        j2 = enc.encode(p, pi, raw)
        assert np.array_equal(enc.coefficients(img.data_size), O.fdct_quant(img, O.preprocess(img, raw))), csi
        enc.close()
        del j2


@pytest.mark.parametrize("cs_from,cs_to", [(1, 3), (1, 2), (1, 4), (3, 1), (2, 1), (4, 1), (1, 1)])
def test_exhaustive_colour_transform_fused(O, G, gpu_lib, cs_from, cs_to):
    """The fused kernels evaluate the integer colour transforms of src/gpujpeg_colorspace.h in fp32 (v_pk_fma_f32 +
    v_cvt_pk_u8_f32): all 2^24 input triples of every instantiated matrix must give the oracle's bytes."""
    import torch
    n = 1 << 24
    v = np.arange(n, dtype=np.uint32)
    triples = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8)
    img = oracle_image(O, ("x", 4096, 4096, 1, cs_from, 90, 36, 0, None, cs_to))  # the oracle's preprocessor = the transform per pixel
    want = O.preprocess(img, triples.reshape(-1)).reshape(3, n).T
    d_in = torch.from_numpy(triples.reshape(-1)).cuda()
    d_out = torch.empty(3 * n, dtype=torch.uint8, device="cuda")
    hooks = C.CDLL(os.path.join(os.path.dirname(G.PRODUCT_LIB), "libgj_testhooks.so"))  # test-only kernels over the product's device header
    fn = hooks.gj_test_color444
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    assert fn(cs_from, cs_to, d_in.data_ptr(), d_out.data_ptr(), n // 8, None) == 0
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().reshape(3, n)
    for c in range(3):
        bad = np.nonzero(got[c] != want[:, c])[0]
        assert bad.size == 0, (c, bad[:5], triples[bad[:5]], got[c][bad[:5]], want[bad[:5], c])


@pytest.mark.parametrize("name,w,h,restart", [("hd_config1", 1920, 1080, 24), ("4k", 3840, 2160, -1), ("8k", 7680, 4320, -1)])
def test_full_size_rgb_bit_exact(O, G, gpu_lib, name, w, h, restart):
    """BASELINE.json configs 1-3 at their full sizes, bit-exact against the oracle."""
    raw = natural_image(w, h, 3, seed=w)
    case = (name, w, h, 1, 1, 75, restart, 0, None, 3)
    want = O.encode(oracle_image(O, case), raw)
    p, pi = api_params(gpu_lib, G, case)
    enc = G.Encoder(gpu_lib)
    jpeg = enc.encode(p, pi, raw)
    assert np.array_equal(jpeg, want)
    assert np.array_equal(enc.encode(p, pi, raw), jpeg), "encoding is deterministic"
    px, _ = G.Decoder(gpu_lib).decode(jpeg)
    assert np.array_equal(px, O.decode(want)[0])
    assert psnr(px, raw) > 30.0


def test_16k_422_interleaved_q90(O, G, gpu_lib):
    """BASELINE.json config 4 at full size: 15360x8640 YCbCr 4:2:2 packed (UYVY), interleaved, q90 -- every byte of the stream and every
    decoded sample against the oracle (k_encode_uyvy422 / the interleaved sub-sequence decoder / k_idct_fused_uyvy422 on 4.1 M blocks,
    172 800 restart segments)."""
    w, h = 15360, 8640
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:h // 8, 0:w // 8]
    base = (128 + 90 * np.sin(xx / 31.0) * np.cos(yy / 17.0)).astype(np.float32)
    y = np.kron(base, np.ones((8, 8), np.float32)) + rng.normal(0, 4, (h, w)).astype(np.float32)
    raw = np.empty((h, w, 2), np.uint8)
    raw[:, :, 1] = np.clip(y, 0, 255)
    raw[:, 0::2, 0] = (110 + 20 * np.sin(xx / 9.0)).astype(np.uint8).repeat(8, 0).repeat(4, 1)
    raw[:, 1::2, 0] = (150 + 20 * np.cos(yy / 7.0)).astype(np.uint8).repeat(8, 0).repeat(4, 1)
    raw = raw.reshape(-1)
    del y, base
    case = ("16k", w, h, 3, 3, 90, -1, 1, None, 3)
    want = O.encode(oracle_image(O, case), raw)
    p, pi = api_params(gpu_lib, G, case)
    enc = G.Encoder(gpu_lib)
    jpeg = enc.encode(p, pi, raw)
    assert np.array_equal(jpeg, want), "16K 4:2:2 stream differs from the oracle"
    assert np.array_equal(enc.encode(p, pi, raw), jpeg), "encoding is deterministic"
    enc.close()
    dec = G.Decoder(gpu_lib)
    dec.set_output_format(3, 3)
    px, info = dec.decode(jpeg)
    assert (info.width, info.height, info.pixel_format) == (w, h, 3)
    assert np.array_equal(px, O.decode(want, 3, 3)[0]), "16K 4:2:2 decoded samples differ from the oracle"
    assert psnr(px, raw) > 38.0
    dec.close()


def test_16k_rgb_vs_oracle(O, G, gpu_lib):
    """The largest RGB frame of the BASELINE configurations, 15360x8640 q75, takes token mode by itself: stream and decoded samples
    against the oracle at full size."""
    w, h = 15360, 8640
    rng = np.random.default_rng(16)
    yy, xx = np.mgrid[0:h // 16, 0:w // 16]
    base = np.stack([128 + 90 * np.sin(xx / 23.0) * np.cos(yy / 17.0), xx * 255.0 / (w // 16), yy * 255.0 / (h // 16)], -1).astype(np.float32)
    img = np.kron(base, np.ones((16, 16, 1), np.float32))
    img += rng.normal(0, 3, (h, w, 1)).astype(np.float32)
    raw = np.clip(img, 0, 255).astype(np.uint8).reshape(-1)
    del img, base
    case = ("16k", w, h, 1, 1, 75, -1, 0, None, 3)
    want = O.encode(oracle_image(O, case), raw)
    p, pi = api_params(gpu_lib, G, case)
    enc = G.Encoder(gpu_lib)
    assert np.array_equal(enc.encode(p, pi, raw), want)
    enc.close()
    dec = G.Decoder(gpu_lib)
    px = dec.decode(want)[0]
    dec.close()
    assert np.array_equal(px, O.decode(want)[0])


@pytest.mark.parametrize("pattern", ["noise", "gradient"])
def test_8k_tst_patterns_vs_oracle(O, G, gpu_lib, pattern):
    """The reference's own synthetic patterns (src/utils/image_delegate.c:562-603, through gpujpeg_image_load_from_file of a .tst name) at
    8K: worst and best case for the entropy coder, bit-exact against the oracle in both directions."""
    import bench
    import torch
    w, h = 7680, 4320
    raw = bench.synth_frame(gpu_lib, w, h, pattern, 12345, torch.device("cuda", 0)).cpu().numpy().reshape(-1)
    if pattern == "noise":
        assert np.array_equal(raw[:64], O.noise(64, seed=12345)), "the product's .tst LCG and the oracle's disagree"
    case = ("8k", w, h, 1, 1, 75, -1, 0, None, 3)
    want = O.encode(oracle_image(O, case), raw)
    p, pi = api_params(gpu_lib, G, case)
    enc = G.Encoder(gpu_lib)
    assert np.array_equal(enc.encode(p, pi, raw), want)
    enc.close()
    dec = G.Decoder(gpu_lib)
    px = dec.decode(want)[0]
    dec.close()
    assert np.array_equal(px, O.decode(want)[0])


def test_batch_frames_vs_oracle(O, G, gpu_lib):
    """BASELINE.json config 5: eight of the 256 frames of the 4K batch (seed 12345 + i, bench.py's generator), stream and decoded
    samples against the oracle."""
    import bench
    import torch
    w, h = 3840, 2160
    case = ("4k", w, h, 1, 1, 75, -1, 0, None, 3)
    p, pi = api_params(gpu_lib, G, case)
    enc, dec = G.Encoder(gpu_lib), G.Decoder(gpu_lib)
    img = oracle_image(O, case)
    for i in (0, 1, 37, 64, 101, 128, 200, 255):
        raw = bench.synth_frame(gpu_lib, w, h, "natural", 12345 + i, torch.device("cuda", 0)).cpu().numpy().reshape(-1)
        want = O.encode(img, raw)
        assert np.array_equal(enc.encode(p, pi, raw), want), i
        assert np.array_equal(dec.decode(want)[0], O.decode(want)[0]), i
    enc.close()
    dec.close()


def test_reconfiguration_and_reuse(O, G, gpu_lib):
    """One encoder/decoder across changing sizes, qualities and layouts (test/regression/run_tests.sh:28-55)."""
    enc, dec = G.Encoder(gpu_lib), G.Decoder(gpu_lib)
    for (w, h, q, il, ri) in [(64, 64, 75, 0, 4), (1119, 561, 75, 0, -1), (64, 64, 75, 0, 4), (640, 480, 30, 1, 7), (640, 480, 95, 1, 7), (16, 16, 50, 0, 0)]:
        raw = O.noise(w * h * 3, seed=w + q)
        case = ("x", w, h, 1, 1, q, ri, il, None, 3)
        p, pi = api_params(gpu_lib, G, case)
        want = O.encode(oracle_image(O, case), raw)
        assert np.array_equal(enc.encode(p, pi, raw), want), (w, h, q, il, ri)
        assert np.array_equal(dec.decode(want)[0], O.decode(want)[0]), (w, h, q, il, ri)


def test_zero_image_round_trip(O, G, gpu_lib):
    """All-zero image with restart interval 1 decodes to exactly zero (test/regression/run_tests.sh:11-25)."""
    w, h = 256, 128
    raw = np.zeros(w * h * 3, np.uint8)
    p, pi = api_params(gpu_lib, G, ("z", w, h, 1, 1, 75, 1, 0, None, 3))
    jpeg = G.Encoder(gpu_lib).encode(p, pi, raw)
    px, _ = G.Decoder(gpu_lib).decode(jpeg)
    assert psnr(px, raw) >= 50.0


def test_random_psnr_floors(O, G, gpu_lib):
    """The reference's own regression floors on seeded-random 1119x561 images (test/regression/run_tests.sh:116-151)."""
    w, h = 1119, 561
    for pf, cs, q, floor in [(1, 1, 75, 22.0), (0, 3, 75, 28.4)]:
        raw = O.noise(O.raw_size(w, h, pf), seed=12345)
        p, pi = api_params(gpu_lib, G, ("r", w, h, pf, cs, q, -1, 0, None, 3))
        jpeg = G.Encoder(gpu_lib).encode(p, pi, raw)
        dec = G.Decoder(gpu_lib)
        dec.set_output_format(cs, pf)
        px, _ = dec.decode(jpeg)
        assert psnr(px, raw) >= floor


def test_width_padding(O, G, gpu_lib):
    w, h, pad = 100, 37, 12
    raw = O.noise((w * 3 + pad) * h, seed=5)
    case = ("pad", w, h, 1, 1, 75, 5, 0, None, 3)
    p, pi = api_params(gpu_lib, G, case)
    pi.width_padding = pad
    img = O.make_image(w, h, restart_interval=5, width_padding=pad)
    # the API counts the padding in PIXELS when it sizes the caller's buffer ((width + width_padding) * height * bpp,
    # src/gpujpeg_common.c:1188) and in BYTES when it walks the rows: the buffer a caller hands over has to be the larger of the two
    buf = np.zeros(gpu_lib.L.gpujpeg_image_calculate_size(C.byref(pi)), np.uint8)
    buf[: raw.size] = raw
    for fused in (True, False):
        enc = G.Encoder(gpu_lib)
        enc.set_fused(fused)
        assert np.array_equal(enc.encode(p, pi, buf), O.encode(img, raw))


def test_encoder_path_changes_between_frames(O, G, gpu_lib):
    """One encoder, alternating between the tile path (k_encode_* + k_gather) and the coefficient-plane paths (fused path off,
    flipped input, kept coefficients) from frame to frame: k_gather's two sets of group totals alternate only between tile-path
    calls (a plane-path call used to flip the set and the next tile-path call wrote a stream of the wrong size)."""
    w, h = 160, 96
    raw = (O.gradient(w, h, 3).astype(np.int32) + O.noise(w * h * 3, seed=11) % 24).clip(0, 255).astype(np.uint8).reshape(-1)
    case = ("paths", w, h, 1, 1, 75, 5, 0, None, 3)
    p, pi = api_params(gpu_lib, G, case)
    want = O.encode(O.make_image(w, h, restart_interval=5), raw)
    flipped_want = None
    enc = G.Encoder(gpu_lib)
    seq = ["tile", "flip", "tile", "tile", "generic", "tile", "keep", "flip", "tile", "tile"]
    for step, kind in enumerate(seq):
        enc.set_fused(kind != "generic")
        enc.keep_coefficients(kind == "keep")
        assert enc.set_option("enc_opt_flipped", "1" if kind == "flip" else "0") == 0
        got = enc.encode(p, pi, raw)
        if kind == "flip":
            if flipped_want is None:
                flipped_want = got.copy()
            assert np.array_equal(got, flipped_want), f"step {step} ({kind})"
        else:
            assert np.array_equal(got, want), f"step {step} ({kind}): {got.size} B against {want.size} B"


# ---- entropy decoder variants: sub-sequence parallel kernel (default), lane-per-segment kernel, and the hand-over of
# ---- segments that do not fit the LDS stage
ENTROPY_CASES = [
    # name, w, h, quality, restart, interleaved, subsampling, noise?
    ("long_segments_noise_q100", 256, 256, 100, 200, 0, None, True),      # segments of ~35 KB: every full one goes to the serial kernel
    ("mixed_noise_q100_r40", 320, 200, 100, 40, 0, None, True),           # ~7 KB segments next to short ones at the row ends
    ("interleaved_420_natural", 400, 300, 85, 3, 1, [(2, 2), (1, 1), (1, 1)], False),
    ("interleaved_444_noise", 200, 120, 90, 7, 1, None, True),
    ("interleaved_long_noise", 128, 128, 100, 60, 1, None, True),         # interleaved segments of ~30 KB: walked by one lane each
    ("restart0_natural", 512, 384, 85, 0, 0, None, False),                # one segment per scan, tens of KB: decoded piece by piece
    ("restart0_interleaved_420", 512, 384, 85, 0, 1, [(2, 2), (1, 1), (1, 1)], False),
    ("restart0_flat", 640, 480, 75, 0, 0, None, None),                     # flat image: thousands of 4-6 bit blocks per piece
    ("tiny_segments_r1", 320, 64, 30, 1, 0, None, False),                  # one block per segment
    ("wide_natural_auto", 1920, 136, 75, -1, 0, None, False),
]


@pytest.mark.parametrize("mode", ["par", "serial", "seq"])
@pytest.mark.parametrize("ec", ENTROPY_CASES, ids=[c[0] for c in ENTROPY_CASES])
def test_entropy_decoder_variants(O, G, gpu_lib, ec, mode, monkeypatch):
    """The three entropy decoders (sub-sequence parallel, lane per segment over stream windows, lane per segment over an LDS stage --
    the last one falls back by itself when a segment does not fit its stage) give the oracle's coefficients."""
    name, w, h, q, ri, il, ss, noisy = ec
    case = (name, w, h, 1, 1, q, ri, il, ss, 3)
    raw = O.noise(w * h * 3, seed=w + h) if noisy else (np.full(w * h * 3, 77, np.uint8) if noisy is None else natural_image(w, h, 3, seed=q))
    want = O.encode(oracle_image(O, case), raw)
    monkeypatch.delenv("GJ_DEC_ENTROPY", raising=False)
    monkeypatch.delenv("GJ_DEC_SEQ", raising=False)
    if mode == "serial":
        monkeypatch.setenv("GJ_DEC_ENTROPY", "serial")
    elif mode == "seq":
        monkeypatch.setenv("GJ_DEC_SEQ", "1")
    dec = G.Decoder(gpu_lib)
    dec.keep_coefficients()
    px, _ = dec.decode(want)
    s = O.parse(want)
    assert np.array_equal(dec.coefficients(s.img.data_size), O.huffman_decode(s, want)), "entropy decoder differs"
    O.lib().gjo_stream_free(C.byref(s))
    assert np.array_equal(px, O.decode(want)[0])
    dec.close()


def test_decoder_reuse_without_clearing(O, G, gpu_lib):
    """The IDCT leaves the coefficient planes zeroed for the next call: alternate streams of the same and of different
    geometry through one decoder and check every result."""
    streams = []
    for i, (w, h, q) in enumerate([(320, 240, 75), (320, 240, 20), (320, 240, 95), (160, 96, 75), (320, 240, 75)]):
        case = ("r", w, h, 1, 1, q, -1, 0, None, 3)
        jpeg = O.encode(oracle_image(O, case), natural_image(w, h, 3, seed=i))
        streams.append((jpeg, O.decode(jpeg)[0]))
    dec = G.Decoder(gpu_lib)
    for fused in (True, False):
        dec.set_fused(fused)
        for jpeg, want_px in streams + streams[::-1]:
            px, _ = dec.decode(jpeg)
            assert np.array_equal(px, want_px)
    dec.close()


@pytest.mark.parametrize("w,h,restart", [(648, 50, 6), (322, 77, 1), (1928, 24, 13), (640, 64, 64), (640, 64, 65), (16, 8, 3), (4096, 40, 2)])
def test_packed_422_whole_frame_encoder(O, G, gpu_lib, w, h, restart):
    """k_encode_uyvy422 (pixels -> segment streams for interleaved packed 4:2:2): 16-byte aligned and unaligned pitches,
    partial MCUs on both edges, restart intervals up to the 256-block tile and one beyond it (generic coder); bytes equal
    the oracle's (src/gpujpeg_huffman_gpu_encoder.cu MCU order Y0 Y1 Cb Cr, DC prediction per component)."""
    raw = O.noise(O.raw_size(w, h, 3), seed=w + restart)
    for q in (90, 35):
        case = ("x", w, h, 3, 3, q, restart, 1, None, 3)
        want = O.encode(oracle_image(O, case), raw)
        p, pi = api_params(gpu_lib, G, case)
        enc = G.Encoder(gpu_lib)
        assert np.array_equal(enc.encode(p, pi, raw), want), (w, h, restart, q)
        enc.keep_coefficients()  # the two-kernel path with coefficient planes
        assert np.array_equal(enc.encode(p, pi, raw), want), (w, h, restart, q, "planes")
        enc.close()


# name, w, h, pixel format, colour space, quality, restart, interleaved, subsampling, noise?
TAIL_CASES = [
    ("rgb_many_tiles", 1024, 520, 1, 1, 75, -1, 0, None, False),          # k_encode_rgb444: 3 x 33 tile streams, chroma behind all luminance
    ("rgb_noise_q100_windows", 512, 264, 1, 1, 100, 32, 0, None, True),   # tile streams of ~25 KB: several LDS windows, bytes stored one by one
    ("rgb_noise_q90", 640, 368, 1, 1, 90, -1, 0, None, True),             # 0xFF bytes in most segments
    ("rgb_short_last_tile", 1000, 200, 1, 1, 60, 7, 0, None, False),      # the scan's last tile is a partial one
    ("uyvy_il", 1288, 240, 3, 3, 90, -1, 1, None, True),                  # k_encode_uyvy422: one scan
    ("rgb_420_il", 644, 482, 1, 1, 75, -1, 1, [(2, 2), (1, 1), (1, 1)], False),   # k_encode_blocks, interleaved
    ("rgb_422_nonil", 800, 300, 1, 1, 80, 5, 0, [(2, 1), (1, 1), (1, 1)], True),  # k_encode_blocks: scans with different tile counts
    ("planar420", 642, 482, 5, 3, 75, -1, 0, None, True),
    ("gray", 999, 333, 0, 3, 75, 6, 0, None, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("tc", TAIL_CASES, ids=[c[0] for c in TAIL_CASES])
def test_encoder_tiles_and_gather(O, G, gpu_lib, tc, monkeypatch):
    """k_encode_* leave the unstuffed stream of every tile and its size in the file; k_gather (one wave per tile stream) places, stuffs
    and marks them: the file's bytes must be the oracle's (replaces src/gpujpeg_huffman_gpu_encoder.cu:417-613 and the host stitching of
    src/gpujpeg_encoder.c:567-629), with and without the APP13 index, three times in a row on the same coder (the group totals
    alternate between two sets that k_gather clears)."""
    name, w, h, pf, cs, q, restart, il, sub, noisy = tc
    case = (name, w, h, pf, cs, q, restart, il, sub, 3)
    comps = {0: 1, 1: 3}.get(pf)
    raw = natural_image(w, h, comps, seed=w) if comps and not noisy else O.noise(O.raw_size(w, h, pf), seed=w * 7 + h)
    # (k_encode_rgb444 codes a small frame one component per workgroup, a large one all three in one: GJ_ENC_SPLIT moves the limit -- both ways here;
    #  and the last tiles of a frame larger than the GPU one component per workgroup behind the whole ones: GJ_ENC_TAIL reaches that mixture with a small frame)
    for split, tail in ((None, None), ("0", None), ("100000", None), ("0", "1"), ("0", "2"), ("0", "5")):
        for var, val in (("GJ_ENC_SPLIT", split), ("GJ_ENC_TAIL", tail)):
            if val is None:
                monkeypatch.delenv(var, raising=False)
            else:
                monkeypatch.setenv(var, val)
        enc = G.Encoder(gpu_lib)  # (reads the switches)
        for seg_info in (0, 1, 0):
            want = O.encode(oracle_image(O, case, segment_info=seg_info), raw)
            p, pi = api_params(gpu_lib, G, case, segment_info=seg_info)
            got = enc.encode(p, pi, raw)
            assert got.size == want.size and np.array_equal(got, want), (name, seg_info, split, tail, got.size, want.size)
        enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [101, 103, 204, 401, 802, 1601, 1604])
def test_marker_scan_shapes(O, G, gpu_lib, shape, monkeypatch):
    """The device's segment table (k_marker_scan + k_marker_table, replaces the host walk of src/gpujpeg_reader.c:1039-1155) for every
    shape of the scan's workgroups -- 4 KB pieces per round x rounds, GJ_SCAN_SHAPE --, which the stream's size chooses otherwise (an 8K
    frame: 8 pieces, config 4: 16 pieces x 3 rounds): three scans whose SOS markers fall into different workgroups, one interleaved
    scan, segments of a few bytes (several restart markers in a lane's 16 bytes), a stream without restart markers."""
    monkeypatch.setenv("GJ_SCAN_SHAPE", str(shape))
    cases = [("a", 640, 368, 1, 1, 90, -1, 0, None, 3), ("b", 322, 250, 3, 3, 90, -1, 1, None, 3), ("c", 320, 64, 1, 1, 30, 1, 0, None, 3),
             ("d", 200, 120, 1, 1, 75, 0, 0, None, 3), ("e", 1024, 520, 1, 1, 95, 5, 0, None, 3)]
    dec = G.Decoder(gpu_lib)  # (reads the switch)
    for case in cases:
        raw = O.noise(O.raw_size(case[1], case[2], case[3]), seed=case[1])
        jpeg = O.encode(oracle_image(O, case), raw)
        want = O.decode(jpeg, 3, 3)[0] if case[3] == 3 else O.decode(jpeg)[0]
        dec.set_output_format(3, 3) if case[3] == 3 else dec.set_output_format(1, 1)  # (colour space, pixel format)
        px, _ = dec.decode(jpeg)
        assert np.array_equal(px, want), (shape, case[0])
    dec.close()


TOKEN_CASES = [
    # name, w, h, quality, restart, noise? (non-interleaved RGB 4:4:4: the configurations the token-fed IDCT serves)
    ("natural_auto", 1920, 136, 75, -1, False),
    ("natural_q90_odd", 1119, 561, 90, 12, False),
    ("noise_q75", 640, 368, 75, -1, True),              # ~30 coefficients per block: more tokens per wave than the LDS stage holds
    ("noise_q100_r40", 320, 200, 100, 40, True),        # long and short segments mixed: records of both kinds in one wave
    ("long_segments_noise_q100", 256, 256, 100, 200, True),
    ("restart0_natural", 512, 384, 85, 0, False),       # every block arrives through the planes
    ("tiny_segments_r1", 320, 64, 30, 1, False),
    ("flat", 640, 480, 75, 36, None),                   # blocks without any token
    ("one_block", 8, 8, 75, 4, True),
]


@pytest.mark.parametrize("tc", TOKEN_CASES, ids=[c[0] for c in TOKEN_CASES])
def test_token_mode_decoder(O, G, gpu_lib, tc, monkeypatch):
    """Token mode (entropy decoder -> dense token array + block records -> k_idct_tok_rgb444) is chosen for large frames
    only; forced here on small ones. Pixels equal the oracle's and the plane-mode result, also when the decoder object
    is reused for another stream."""
    name, w, h, q, ri, noisy = tc
    case = (name, w, h, 1, 1, q, ri, 0, None, 3)
    raw = O.noise(w * h * 3, seed=w + h) if noisy else (np.full(w * h * 3, 77, np.uint8) if noisy is None else natural_image(w, h, 3, seed=q))
    jpeg = O.encode(oracle_image(O, case), raw)
    want = O.decode(jpeg)[0]
    other = O.encode(oracle_image(O, ("o", 333, 123, 1, 1, 60, 5, 0, None, 3)), natural_image(333, 123, 3, seed=3))
    monkeypatch.setenv("GJ_DEC_TOKENS", "1")
    dec = G.Decoder(gpu_lib)
    for _ in range(2):
        px, _ = dec.decode(jpeg)
        assert np.array_equal(px, want)
        assert np.array_equal(dec.decode(other)[0], O.decode(other)[0])
    dec.close()
    monkeypatch.setenv("GJ_DEC_SEQ", "1")  # token mode through the lane-per-segment kernel
    dec = G.Decoder(gpu_lib)
    assert np.array_equal(dec.decode(jpeg)[0], want)
    dec.close()
    monkeypatch.delenv("GJ_DEC_SEQ")
    monkeypatch.setenv("GJ_DEC_NO_TOKENS", "1")  # (the switches are read when a decoder is created)
    dec = G.Decoder(gpu_lib)
    assert np.array_equal(dec.decode(jpeg)[0], want)
    dec.close()


FOLD_CASES = [
    # name, w, h, quality, restart, pattern -- non-interleaved RGB 4:4:4 through k_huffman_decode_tok on the speculative path
    ("natural_auto", 1920, 136, 75, -1, "natural"),
    ("natural_q90_odd", 1119, 561, 90, 12, "natural"),
    ("noise_q75", 640, 368, 75, -1, "noise"),
    ("tiny_segments_r1", 320, 64, 30, 1, "natural"),  # ~1000 restart markers per scanning workgroup
    ("flat", 640, 480, 75, 36, "flat"),
    ("one_segment_per_scan", 64, 32, 75, 40, "natural"),  # no restart marker at all: every scan is its own last segment
    ("long_segments_q100_noise", 640, 368, 100, 120, "noise"),  # 17 KB segments: decoded in pieces cut from the table, so the fold is refused up front (it was decoded twice, every frame)
    # interleaved scans (one SOS, every segment in it) through k_huffman_decode_par<interleaved>: name, w, h, quality, restart, pattern, sampling
    ("il_444", 640, 368, 75, -1, "natural", None),
    ("il_420_odd", 645, 483, 85, 7, "natural", [(2, 2), (1, 1), (1, 1)]),
    ("il_422_noise", 800, 304, 75, 5, "noise", [(2, 1), (1, 1), (1, 1)]),
]


@pytest.mark.parametrize("kernel", ["tok", "par"])
@pytest.mark.parametrize("fc", FOLD_CASES, ids=[c[0] for c in FOLD_CASES])
def test_token_decoder_without_the_table_launch(O, G, gpu_lib, fc, monkeypatch, kernel):
    """Frames of a sequence (one header) on one decoder: from the second one on the launch is speculative and the token decoder derives its batches'
    segment table from the marker scan's records itself -- no k_marker_table launch (gj_scan_deferred, round 5). Pixels equal the oracle's; streams
    that are NOT the complete, regular stream the geometry describes (restart markers out of sequence, missing, surplus; no EOI; a stranger's header)
    are noticed on the device and decoded again the careful way, with the result a fresh decoder gives."""
    name, w, h, q, ri, pattern = fc[:6]
    interleaved = len(fc) > 6
    if interleaved and kernel == "tok":
        pytest.skip("interleaved scans do not take the token decoder of non-interleaved frames")
    case = (name, w, h, 1, 1, q, ri, 1 if interleaved else 0, fc[6] if interleaved else None, 3)
    img = oracle_image(O, case)

    def frame(seed):
        if pattern == "noise":
            return O.noise(w * h * 3, seed=seed)
        if pattern == "flat":
            return np.full(w * h * 3, 60 + seed % 100, np.uint8)
        return natural_image(w, h, 3, seed=seed)

    streams = [O.encode(img, frame(40 + f)) for f in range(4)]
    if kernel == "tok":
        monkeypatch.setenv("GJ_DEC_TOKENS", "1")  # k_huffman_decode_tok; without it frames of this size take k_huffman_decode_par (planes), which folds the table launch the same way
    dec = G.Decoder(gpu_lib)
    for rnd in range(2):
        for f, jpeg in enumerate(streams):
            assert np.array_equal(dec.decode(jpeg)[0], O.decode(jpeg)[0]), (rnd, f)
    spec, folded, again = dec.path_counters()
    # (a stream whose three scan-ending markers all lie in ONE scanning workgroup's part -- a few KB: the last two cases -- is walked by the host and
    # never launches speculatively: a scanning workgroup's record holds two markers that are no restart markers)
    assert spec == (0 if name in ("tiny_segments_r1", "one_segment_per_scan") else 7) and again == 0, (spec, folded, again)
    if name == "long_segments_q100_noise":  # segments beyond the LDS stage are cut into pieces from the table: speculative, with the table launch, decoded ONCE
        assert folded == 0, (spec, folded, again)
    else:
        assert folded == spec, "every speculative launch of this geometry does without the table kernel"
    # ---- streams the geometry does not describe, on the speculative path
    jpeg = streams[1]
    pos = [i for i in range(len(jpeg) - 1) if jpeg[i] == 0xFF and 0xD0 <= jpeg[i + 1] <= 0xD7]
    bad = []
    if len(pos) >= 4:
        a = jpeg.copy(); a[pos[1] + 1], a[pos[2] + 1] = a[pos[2] + 1], a[pos[1] + 1]; bad.append(("swapped numbers", a))
        b = np.delete(jpeg, [pos[len(pos) // 2], pos[len(pos) // 2] + 1]); bad.append(("missing marker", b))
        c = np.insert(jpeg, pos[-1], [0xFF, 0xD0 + ((jpeg[pos[-1] + 1] - 0xD0 + 7) % 8)]); bad.append(("surplus marker", c))
    bad.append(("no EOI", jpeg[:-2].copy()))
    other = O.encode(oracle_image(O, ("o", w, h, 1, 1, max(10, q - 20), ri, case[7], case[8], 3)), frame(7))  # same dimensions, other tables: a stranger's header
    bad.append(("other header", other))
    for what, b in bad:
        fresh = G.Decoder(gpu_lib)
        try:
            want = fresh.decode(b)[0]
        except RuntimeError:
            want = None
        fresh.close()
        assert np.array_equal(dec.decode(jpeg)[0], O.decode(jpeg)[0])  # (the cached header is this sequence's again)
        try:
            got = dec.decode(b)[0]
        except RuntimeError:
            got = None
        assert (got is None) == (want is None), what
        if want is not None:
            assert np.array_equal(got, want), what
    assert np.array_equal(dec.decode(jpeg)[0], O.decode(jpeg)[0])
    dec.close()


def test_token_mode_longer_sub_sequences(O, G, gpu_lib, monkeypatch):
    """k_huffman_decode_tok cuts a group into sub-sequences of 17..20 bytes instead of 16 when that saves it a whole pass of its 256 lanes
    (an 8K frame's luminance batches). Forced here on a small frame: batches of GJ_DEC_G segments whose groups hold a little more than
    256 (and a little more than 512) sub-sequences of 16 bytes."""
    w, h, q, ri = 1024, 512, 92, 8
    case = ("longsub", w, h, 1, 1, q, ri, 0, None, 3)
    raw = natural_image(w, h, 3, seed=17)
    jpeg = O.encode(oracle_image(O, case), raw)
    want = O.decode(jpeg)[0]
    nseg = 3 * ((w // 8) * (h // 8) // ri)
    monkeypatch.setenv("GJ_DEC_TOKENS", "1")
    for group_bytes in (4500, 4900, 8600):  # (6 - 13 of the ~50 groups of each run take longer sub-sequences: GJ_TOK_STATS build, tools/tok_lane_stats.py)
        monkeypatch.setenv("GJ_DEC_G", str(max(1, min(64, round(group_bytes * nseg / jpeg.size)))))
        dec = G.Decoder(gpu_lib)
        assert np.array_equal(dec.decode(jpeg)[0], want)
        dec.close()


def test_token_mode_tiny_segments(O, G, gpu_lib, monkeypatch):
    """The cooperative copy of k_huffman_decode_tok classifies the stream four bytes at a time and places what it keeps around the markers
    of a dword: restart intervals of one and two blocks give segments of one to three bytes -- two markers in one dword, markers that
    straddle dwords and lanes, 0xFF data bytes with their stuffing next to markers (noise)."""
    monkeypatch.setenv("GJ_DEC_TOKENS", "1")
    for w, h, q, ri, kind in ((256, 64, 75, 1, "flat"), (256, 64, 30, 1, "natural"), (264, 40, 75, 2, "natural"), (128, 64, 95, 1, "noise"), (200, 48, 100, 3, "noise")):
        raw = np.full(w * h * 3, 200, np.uint8) if kind == "flat" else O.noise(w * h * 3, seed=ri + q) if kind == "noise" else natural_image(w, h, 3, seed=q)
        jpeg = O.encode(oracle_image(O, ("tiny", w, h, 1, 1, q, ri, 0, None, 3)), raw)
        want = O.decode(jpeg)[0]
        for g in ("64", "7"):
            monkeypatch.setenv("GJ_DEC_G", g)
            dec = G.Decoder(gpu_lib)
            assert np.array_equal(dec.decode(jpeg)[0], want), (w, h, q, ri, kind, g)
            dec.close()


def test_token_mode_damaged_streams(O, G, gpu_lib, monkeypatch):
    """Token mode on damaged input: flipped bytes and truncation must neither fault nor hang (records nobody wrote, token
    counts that no longer match)."""
    monkeypatch.setenv("GJ_DEC_TOKENS", "1")
    w, h = 640, 368
    jpeg = O.encode(oracle_image(O, ("d", w, h, 1, 1, 75, -1, 0, None, 3)), natural_image(w, h, 3, seed=5))
    rng = np.random.default_rng(11)
    dec = G.Decoder(gpu_lib)
    good = dec.decode(jpeg)[0]
    for trial in range(10):
        bad = jpeg.copy()
        if trial % 3 == 2:
            bad = bad[: int(bad.size * rng.uniform(0.3, 0.95))]
        else:
            lo = 700
            for i in rng.integers(lo, bad.size - 2, size=1 + trial):
                bad[i] = rng.integers(0, 256)
        try:
            dec.decode(bad)
        except Exception:
            pass
        assert np.array_equal(dec.decode(jpeg)[0], good), trial  # the decoder is intact afterwards
    dec.close()


@pytest.mark.parametrize("layout", ["rgb444", "uyvy422"])
def test_damaged_streams_token_mode_equals_plane_mode(O, G, gpu_lib, layout, monkeypatch):
    """What a damaged stream decodes to must not depend on the path the frame's size chooses: a coefficient whose run carries it past
    the end of its block is dropped by every entropy decoder -- in token mode it becomes a token on position 0, which the IDCT
    overwrites with the DC term --, as the reference's GPU decoder does (src/gpujpeg_huffman_gpu_decoder.cu:370). Bytes inside the
    entropy-coded data are replaced (never by 0xFF: the marker structure stays), token mode and plane mode decode the same samples."""
    w, h = (640, 368) if layout == "rgb444" else (1288, 120)
    if layout == "rgb444":
        case, fmt = ("d", w, h, 1, 1, 90, -1, 0, None, 3), (1, 1)
        raw = natural_image(w, h, 3, seed=5)
    else:
        case, fmt = ("d", w, h, 3, 3, 90, -1, 1, None, 3), (3, 3)
        raw = O.noise(O.raw_size(w, h, 3), seed=9)
    jpeg = O.encode(oracle_image(O, case), raw)
    rng = np.random.default_rng(23)
    decs = []
    for env in ({"GJ_DEC_TOKENS": "1"}, {"GJ_DEC_TOKENS": "1", "GJ_DEC_SEQ": "1"}, {"GJ_DEC_NO_TOKENS": "1"}):
        for k in ("GJ_DEC_TOKENS", "GJ_DEC_SEQ", "GJ_DEC_NO_TOKENS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        d = G.Decoder(gpu_lib)
        d.set_output_format(*fmt)
        decs.append(d)
    for trial in range(12):
        bad = jpeg.copy()
        for i in rng.integers(800, bad.size - 4, size=1 + trial):
            if bad[i] != 0xFF and bad[i - 1] != 0xFF:
                bad[i] = rng.integers(0, 255)
        outs = [d.decode(bad)[0] for d in decs]
        assert np.array_equal(outs[0], outs[2]), (layout, trial, "token mode vs planes")
        assert np.array_equal(outs[1], outs[2]), (layout, trial, "lane-per-segment token mode vs planes")
    for d in decs:
        d.close()


TOKEN_422_CASES = [
    # name, w, h, quality, restart, noise?  (packed 4:2:2 in and out, interleaved scan: k_idct_tok_uyvy422)
    ("natural_auto", 1920, 136, 90, -1, False),
    ("odd_edges", 642, 77, 90, 6, False),           # partial MCUs on both edges, pitch not a multiple of 8
    ("aligned_noise", 640, 64, 75, 5, True),        # more tokens per wave than the LDS stage holds
    ("restart0", 512, 128, 85, 0, False),           # every block arrives through the planes
    ("r1", 320, 40, 50, 1, False),
    ("one_mcu", 16, 8, 90, 3, True),
    ("long_segments_noise", 640, 64, 95, 40, True),  # segments of 160 blocks, several KB each: the ring decoder refills dozens of times per segment
]


@pytest.mark.parametrize("tc", TOKEN_422_CASES, ids=[c[0] for c in TOKEN_422_CASES])
def test_token_mode_decoder_422(O, G, gpu_lib, tc, monkeypatch):
    name, w, h, q, ri, noisy = tc
    case = (name, w, h, 3, 3, q, ri, 1, None, 3)
    if noisy:
        raw = O.noise(O.raw_size(w, h, 3), seed=w + h)
    else:
        rgb = natural_image(w + (w & 1), h, 3, seed=q).reshape(h, w + (w & 1), 3)
        raw = np.empty((h, w + (w & 1), 2), np.uint8)
        raw[:, :, 1] = rgb[:, :, 0]
        raw[:, 0::2, 0] = rgb[:, 0::2, 1]
        raw[:, 1::2, 0] = rgb[:, 0::2, 2]
        raw = raw.reshape(-1)[: O.raw_size(w, h, 3)].copy()
    jpeg = O.encode(oracle_image(O, case), raw)
    want = O.decode(jpeg, 3, 3)[0]
    monkeypatch.setenv("GJ_DEC_TOKENS", "1")  # (the switches are read when a decoder is created)
    dec = G.Decoder(gpu_lib)
    dec.set_output_format(3, 3)
    for _ in range(2):
        assert np.array_equal(dec.decode(jpeg)[0], want)
    dec.close()
    monkeypatch.setenv("GJ_DEC_SEQ", "1")  # token mode through the lane-per-segment kernel + k_idct_tok_uyvy422 (what BASELINE config 4 takes by itself)
    dec = G.Decoder(gpu_lib)
    dec.set_output_format(3, 3)
    for _ in range(2):
        assert np.array_equal(dec.decode(jpeg)[0], want)
    dec.close()
    monkeypatch.delenv("GJ_DEC_SEQ")
    monkeypatch.setenv("GJ_DEC_NO_TOKENS", "1")
    dec = G.Decoder(gpu_lib)
    dec.set_output_format(3, 3)
    assert np.array_equal(dec.decode(jpeg)[0], want)
    dec.close()





def dense_two_bit_stream(w, h, ri):
    """A baseline JPEG (4:2:2, interleaved, restart interval `ri` MCUs) as an encoder with OPTIMISED Huffman tables may write it: the AC
    symbol (run 0, size 1) has a 1-bit code, so a coefficient costs 2 bits of stream -- the densest stream of non-zero coefficients the
    format allows (every AC coefficient of every block is +-1, DC differences 0). Returns the file's bytes."""
    def seg(marker, payload):
        return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload
    out = bytearray(b"\xff\xd8")
    out += seg(0xDB, bytes([0]) + bytes([1] * 64))                                  # one quantisation table, all ones
    out += seg(0xC0, bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([3, 1, 0x21, 0, 2, 0x11, 0, 3, 0x11, 0]))
    # DC table: symbol 0 -> '0' (1 bit), symbol 1 -> '10'; AC table: 0x01 -> '0' (1 bit), 0x00 (EOB) -> '10', 0xF0 -> '110'
    out += seg(0xC4, bytes([0x00]) + bytes([1, 1] + [0] * 14) + bytes([0, 1]))
    out += seg(0xC4, bytes([0x10]) + bytes([1, 1, 1] + [0] * 13) + bytes([0x01, 0x00, 0xF0]))
    out += seg(0xDD, ri.to_bytes(2, "big"))
    out += seg(0xDA, bytes([3, 1, 0x00, 2, 0x00, 3, 0x00, 0, 63, 0]))
    mcus = ((w + 15) // 16) * ((h + 7) // 8)
    rng = np.random.default_rng(w + h)
    for m0 in range(0, mcus, ri):
        bits = []
        for _ in range(min(ri, mcus - m0) * 4):
            bits.append("0")                                                          # DC difference 0
            signs = rng.integers(0, 2, 63)
            bits.append("".join("0" + ("1" if s else "0") for s in signs))            # 63 x (run 0, size 1) + sign bit: +1 / -1
        b = "".join(bits)
        b += "1" * (-len(b) % 8)
        data = int(b, 2).to_bytes(len(b) // 8, "big")
        out += data.replace(b"\xff", b"\xff\x00")
        if m0 + ri < mcus:
            out += bytes([0xFF, 0xD0 + (m0 // ri) % 8])
    out += b"\xff\xd9"
    return np.frombuffer(bytes(out), np.uint8).copy()


@pytest.mark.parametrize("w,h,ri", [(640, 64, 5), (320, 40, 1), (1024, 72, 2)])
def test_dense_two_bit_tokens_all_decoder_paths(O, G, gpu_lib, w, h, ri, monkeypatch):
    """A foreign file with optimised Huffman tables: coefficients of 2 bits each (a token of the decoder's token mode is assumed to cost
    3 with the standard tables; 4 tokens per stream byte is what its array provides). Every entropy decoder, token and plane mode, must
    give the oracle's samples (src/gpujpeg_huffman_gpu_decoder.cu:287-495 decodes any baseline table)."""
    jpeg = dense_two_bit_stream(w, h, ri)
    want = O.decode(jpeg, 3, 3)[0]
    for env in ({}, {"GJ_DEC_TOKENS": "1"}, {"GJ_DEC_TOKENS": "1", "GJ_DEC_SEQ": "1"}, {"GJ_DEC_NO_TOKENS": "1"}, {"GJ_DEC_SEQ": "1"}, {"GJ_DEC_ENTROPY": "serial"}):
        for k in ("GJ_DEC_TOKENS", "GJ_DEC_SEQ", "GJ_DEC_NO_TOKENS", "GJ_DEC_ENTROPY"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dec = G.Decoder(gpu_lib)
        dec.set_output_format(3, 3)
        for _ in range(2):
            assert np.array_equal(dec.decode(jpeg)[0], want), (w, h, ri, env)
        dec.close()


@pytest.mark.parametrize("seed", range(160))
def test_random_configurations(O, G, gpu_lib, seed):
    """Differential test over the configuration space the fixed CASES only sample: the product's stream and decoded samples must equal
    the oracle's for random pixel formats, colour spaces, chroma samplings, odd sizes, qualities, restart intervals and interleaving
    (exercises k_encode_rgb444 / k_encode_blocks / k_encode_uyvy422 / the generic chain and both entropy decoders at their edges; the
    same cases pin the oracle to the reference in tests/test_oracle_vs_ref.py)."""
    from conftest import random_case, random_raw
    case = random_case(seed)
    raw = random_raw(O, case, seed)
    want = O.encode(oracle_image(O, case), raw)
    p, pi = api_params(gpu_lib, G, case)
    enc = G.Encoder(gpu_lib)
    jpeg = enc.encode(p, pi, raw)
    assert jpeg.size == want.size and np.array_equal(jpeg, want), (case, "stream differs")
    enc.keep_coefficients()  # the kernels that go through the coefficient planes
    assert np.array_equal(enc.encode(p, pi, raw), want), (case, "stream differs (coefficient planes)")
    enc.set_fused(False)     # the generic chain: k_preprocess / k_copy_planes_in, k_dct, k_huffman
    assert np.array_equal(enc.encode(p, pi, raw), want), (case, "stream differs (generic kernels)")
    dec = G.Decoder(gpu_lib)
    px, info = dec.decode(want)
    want_px, _ = O.decode(want)
    assert np.array_equal(px, want_px), (case, "decoded samples differ")
    dec.set_fused(False)
    assert np.array_equal(dec.decode(want)[0], want_px), (case, "decoded samples differ (generic kernels)")
    enc.close()
    dec.close()


@pytest.mark.parametrize("seed", range(40))
def test_random_streams_all_decoder_paths(O, G, gpu_lib, seed, monkeypatch):
    """Random mid-size streams of the two layouts with token-fed IDCT kernels (RGB 4:4:4 non-interleaved, packed 4:2:2 interleaved)
    through every decoder path: token mode, plane mode, the lane-per-segment entropy decoder and the generic kernels -- each must give
    the oracle's samples (batches that straddle scans, ragged last batches, long and empty segments, odd sizes)."""
    rng = np.random.default_rng(4000 + seed)
    uyvy = seed % 3 == 2
    w, h = int(rng.integers(8, 900)), int(rng.integers(8, 500))
    if uyvy:
        w += w & 1
    q = int(rng.choice([5, 30, 60, 75, 85, 92, 100]))
    ri = int(rng.choice([-1, -1, 0, 1, 2, 4, 7, 16, 36, 64, 250]))
    case = (f"r{seed}", w, h, 3 if uyvy else 1, 3 if uyvy else 1, q, ri, 1 if uyvy else 0, None, 3)
    n = O.raw_size(w, h, case[3])
    kind = int(rng.integers(0, 3))
    raw = (natural_image(w, h, 3, seed=seed) if kind == 0 and not uyvy else O.noise(n, seed=seed) if kind == 1 else
           ((np.arange(n, dtype=np.int64) // 5 + (O.noise(n, seed=seed) & 7)) % 256).astype(np.uint8))
    jpeg = O.encode(oracle_image(O, case), raw)
    want = O.decode(jpeg, case[3], case[4])[0] if uyvy else O.decode(jpeg)[0]
    for env in ({"GJ_DEC_TOKENS": "1"}, {"GJ_DEC_NO_TOKENS": "1"}, {"GJ_DEC_ENTROPY": "serial"}, {"GPUJPEG_NO_FUSED": "1"}, {"GJ_DEC_NO_TOKENS": "1", "GJ_DEC_SEQ": "1"},
                {"GJ_DEC_TOKENS": "1", "GJ_DEC_SEQ": "1"}):
        for k in ("GJ_DEC_TOKENS", "GJ_DEC_NO_TOKENS", "GJ_DEC_ENTROPY", "GPUJPEG_NO_FUSED", "GJ_DEC_SEQ"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dec = G.Decoder(gpu_lib)  # (the switches are read when a decoder is created)
        if uyvy:
            dec.set_output_format(3, 3)
        px, _ = dec.decode(jpeg)
        assert np.array_equal(px, want), (case, env)
        dec.close()


# ---- frame batches (include/gpujpeg_amd_ext.h: gpujpeg_amd_encoder_encode_batch / gpujpeg_amd_decoder_decode_batch)
# name, w, h, pixel format, quality, restart, interleaved, subsampling, frames, GJ_DEC_TOKENS, batched launches expected (encoder, decoder)
BATCH_CASES = [
    ("planes_640x480", 640, 480, 1, 75, -1, 0, None, 5, None, True, True),
    ("tokens_640x480", 640, 480, 1, 75, -1, 0, None, 5, "1", True, True),
    ("odd_size_q90", 331, 277, 1, 90, 7, 0, None, 3, None, True, True),
    ("two_chunks_of_small_frames", 64, 64, 1, 50, 4, 0, None, 70, None, True, False),  # (a 2 KB stream has its three scans inside one scanning workgroup: host walk)
    # every layout that goes from pixels to tile streams (k_encode_blocks, k_encode_uyvy422) and back through the sub-sequence decoder's planes and any
    # of the IDCT-side kernels (k_idct_fused_uyvy422; k_idct + k_postprocess / k_copy_planes_out)
    ("interleaved_rgb", 320, 240, 1, 75, -1, 1, None, 3, None, True, True),
    ("planar_420", 320, 240, 5, 75, -1, 0, None, 3, None, True, True),
    ("uyvy_422_interleaved", 322, 150, 3, 90, -1, 1, None, 4, None, True, True),
    ("rgb_to_420_interleaved", 320, 240, 1, 75, 3, 1, [(2, 2), (1, 1), (1, 1)], 3, None, True, True),
    ("gray", 333, 211, 0, 75, -1, 0, None, 3, None, True, True),
    # 65 frames x 256 one-MCU segments in flight: the interleaved scan goes through the ring kernel into tokens and the token-fed 4:2:2 IDCT
    ("uyvy_422_tokens_ring", 256, 128, 3, 90, 1, 1, None, 66, "1", True, True),
    ("restart_0", 160, 120, 1, 75, 0, 0, None, 2, None, False, False),                  # one segment per scan: coefficient planes + k_huffman, frame by frame
]


@pytest.mark.parametrize("bc", BATCH_CASES, ids=[c[0] for c in BATCH_CASES])
def test_frame_batches(O, G, gpu_lib, bc, monkeypatch):
    """N frames of one geometry behind one set of launches (blockIdx.z = frame): every stream equals the oracle's, every decoded frame the
    oracle's decoding of it -- i.e. the batch calls give what frame-at-a-time calls give --, also when the coder objects are reused with
    fewer and more frames, when single calls follow, and for configurations the batched kernels do not cover (coded frame by frame
    inside the call). last_batch() says which way the frames went."""
    name, w, h, pf, q, ri, il, ss, n, tok, enc_batched, dec_batched = bc
    case = (name, w, h, pf, 1 if pf == 1 else 3, q, ri, il, ss, 3)
    img = oracle_image(O, case)
    p, pi = api_params(gpu_lib, G, case)
    size = gpu_lib.image_size(pi)
    if pf == 1:
        frames = np.stack([natural_image(w, h, 3, seed=10 + f) for f in range(n)])
    else:
        frames = np.stack([O.noise(size, seed=20 + f) // 2 + 60 for f in range(n)]).astype(np.uint8)
    want = [O.encode(img, frames[f]) for f in range(n)]
    native = pf == 3  # packed 4:2:2 comes back as packed 4:2:2 (k_idct_fused_uyvy422 / k_idct_tok_uyvy422), the others in the decoder's default format
    want_px = [(O.decode(s, 3, 3) if native else O.decode(s))[0] for s in want]
    if tok:
        monkeypatch.setenv("GJ_DEC_TOKENS", tok)
    enc, dec = G.Encoder(gpu_lib), G.Decoder(gpu_lib)
    if native:
        dec.set_output_format(3, 3)
    if name in ("planes_640x480", "tokens_640x480", "planar_420"):  # several chunks per call: 2 + 2 + 1 frames
        enc.set_batch_chunk(2)
        dec.set_batch_chunk(2)
    if name == "two_chunks_of_small_frames":
        enc.set_batch_chunk(32)
    if name == "uyvy_422_tokens_ring":  # 64 + 1 frames (the ring kernel wants 16 384 segments in flight: 64 frames of 256)
        dec.set_batch_chunk(64)
    for count in (n, 2, n):
        got = enc.encode_batch(p, pi, frames[:count].reshape(-1), count)
        assert enc.last_batch() == ((count, 0) if enc_batched else (0, count))
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), "stream of a batched frame differs from the oracle"
        px, info = dec.decode_batch(got)
        assert (info.width, info.height) == (w, h)
        assert all(np.array_equal(a, b) for a, b in zip(px, want_px)), "pixels of a batched frame differ from the oracle"
        # (host memory in and out: the first frame goes the ordinary way, it is the one that is parsed)
        assert dec.last_batch() == ((count - 1, 1) if dec_batched else (0, count))
    assert np.array_equal(enc.encode(p, pi, frames[1]), want[1])
    assert np.array_equal(dec.decode(want[1])[0], want_px[1])
    enc.close()
    dec.close()


def test_frame_batch_with_strangers(O, G, gpu_lib):
    """A batch is launched on ONE header. A stream with another header (other quality), a stream with a damaged restart marker and a truncated
    stream in the middle of it are found by the per-frame validation and decoded the ordinary way; every frame still equals the oracle's."""
    w, h, n = 640, 480, 6
    case = ("b", w, h, 1, 1, 75, -1, 0, None, 3)
    frames = [natural_image(w, h, 3, seed=40 + f) for f in range(n)]
    streams = [O.encode(oracle_image(O, case), f) for f in frames]
    streams[2] = O.encode(oracle_image(O, ("c", w, h, 1, 1, 50, -1, 0, None, 3)), frames[2])  # other tables, same size
    bad = streams[4].copy()
    pos = [i for i in range(len(bad) - 1) if bad[i] == 0xFF and 0xD0 <= bad[i + 1] <= 0xD7]
    bad[pos[len(pos) // 2] + 1] = 0xD0 + ((int(bad[pos[len(pos) // 2] + 1]) - 0xD0 + 3) & 7)  # a restart marker with the wrong number
    streams[4] = bad
    want = [O.decode(s)[0] for s in streams]
    dec = G.Decoder(gpu_lib)
    px, _ = dec.decode_batch(streams)
    batched, single = dec.last_batch()
    assert batched == 3 and single == 3, (batched, single)  # frames 1, 3, 5 | 0 (parsed), 2, 4
    for f in (0, 1, 2, 3, 5):
        assert np.array_equal(px[f], want[f]), f
    ref = G.Decoder(gpu_lib)
    assert np.array_equal(px[4], ref.decode(streams[4])[0])  # (damaged: whatever the ordinary call makes of it)
    # a frame of other dimensions is not what the caller promised: the call fails, nothing is written over a neighbour's slot or behind the buffer
    small = O.encode(oracle_image(O, ("s", 320, 240, 1, 1, 75, -1, 0, None, 3)), natural_image(320, 240, 3, seed=9))
    big = O.encode(oracle_image(O, ("g", 800, 600, 1, 1, 75, -1, 0, None, 3)), natural_image(800, 600, 3, seed=9))
    for other in (small, big):
        with pytest.raises(RuntimeError):
            dec.decode_batch([streams[0], streams[1], other, streams[3]])
    assert np.array_equal(dec.decode(streams[1])[0], want[1])
    dec.close()
    ref.close()


def test_frame_batch_argument_errors(O, G, gpu_lib):
    """the batch calls refuse what they cannot do (return -1, nothing written behind a caller's buffer): no frames, a stride smaller than a
    frame, an output stride smaller than a decoded frame, a stream that is no JPEG -- and work again afterwards"""
    w, h, n = 320, 240, 3
    case = ("e", w, h, 1, 1, 75, -1, 0, None, 3)
    p, pi = api_params(gpu_lib, G, case)
    frames = np.stack([natural_image(w, h, 3, seed=70 + f) for f in range(n)])
    raw = w * h * 3
    enc, dec = G.Encoder(gpu_lib), G.Decoder(gpu_lib)
    ptrs, sizes = (C.c_void_p * n)(), (C.c_size_t * n)()
    L = gpu_lib.L
    assert L.gpujpeg_amd_encoder_encode_batch(enc.h, C.byref(p), C.byref(pi), frames.ctypes.data, raw, 0, ptrs, sizes) == -1
    assert L.gpujpeg_amd_encoder_encode_batch(enc.h, C.byref(p), C.byref(pi), frames.ctypes.data, raw - 1, n, ptrs, sizes) == -1
    streams = enc.encode_batch(p, pi, frames.reshape(-1), n)
    want = [O.decode(s)[0] for s in streams]
    stride = (max(s.size for s in streams) + 79) & ~15
    buf = np.zeros(stride * n, np.uint8)
    for i, s in enumerate(streams):
        buf[i * stride:i * stride + s.size] = s
    csz = (C.c_size_t * n)(*[s.size for s in streams])
    out = np.full(raw * n + 64, 0xAB, np.uint8)
    opi = G.ImageParameters()
    assert L.gpujpeg_amd_decoder_decode_batch(dec.h, buf.ctypes.data, stride, csz, 0, out.ctypes.data, raw, C.byref(opi)) == -1
    assert L.gpujpeg_amd_decoder_decode_batch(dec.h, buf.ctypes.data, stride, csz, n, out.ctypes.data, raw - 16, C.byref(opi)) == -1
    junk = buf.copy()
    junk[stride:stride + 64] = 0x55  # the second stream is no JPEG any more
    assert L.gpujpeg_amd_decoder_decode_batch(dec.h, junk.ctypes.data, stride, csz, n, out.ctypes.data, raw, C.byref(opi)) == -1
    assert np.all(out[raw * n:] == 0xAB)
    assert L.gpujpeg_amd_decoder_decode_batch(dec.h, buf.ctypes.data, stride, csz, n, out.ctypes.data, raw, C.byref(opi)) == 0
    assert all(np.array_equal(out[i * raw:(i + 1) * raw], want[i]) for i in range(n)) and np.all(out[raw * n:] == 0xAB)
    enc.close()
    dec.close()


def test_frame_batch_device_resident(O, G, gpu_lib):
    """frames, streams and decoded frames in device memory (what bench.py does): once the decoder knows the header EVERY frame goes through the batched
    launches -- none is parsed on the host --, and streams that are not 16 bytes apart in device memory are still decoded (frame by frame)"""
    L = gpu_lib.L
    L.gj_hip_malloc.restype = C.c_void_p
    L.gj_hip_malloc.argtypes = [C.c_size_t]
    L.gj_hip_free.argtypes = [C.c_void_p]
    for f in (L.gj_hip_memcpy_h2d, L.gj_hip_memcpy_d2h, L.gj_hip_memcpy_d2d):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.gj_hip_stream_sync.argtypes = [C.c_void_p]
    w, h, n = 640, 480, 4
    case = ("d", w, h, 1, 1, 75, -1, 0, None, 3)
    p, pi = api_params(gpu_lib, G, case)
    raw = w * h * 3
    frames = np.stack([natural_image(w, h, 3, seed=80 + f) for f in range(n)])
    want = [O.encode(oracle_image(O, case), frames[f]) for f in range(n)]
    want_px = [O.decode(s)[0] for s in want]
    d_frames, d_out = L.gj_hip_malloc(raw * n), L.gj_hip_malloc(raw * n)
    assert d_frames and d_out
    assert L.gj_hip_memcpy_h2d(d_frames, frames.ctypes.data, raw * n, None) == 0 and L.gj_hip_stream_sync(None) == 0
    enc, dec = G.Encoder(gpu_lib), G.Decoder(gpu_lib)
    assert enc.set_option("enc_opt_out", "enc_out_val_device") == 0
    ptrs, sizes = enc.encode_batch_noclone(p, pi, d_frames, n, stride=raw, gpu=True)
    assert enc.last_batch() == (n, 0) and sizes == [s.size for s in want]
    got = np.empty(max(sizes), np.uint8)
    for f in range(n):
        assert L.gj_hip_memcpy_d2h(got.ctypes.data, ptrs[f], sizes[f], None) == 0 and L.gj_hip_stream_sync(None) == 0
        assert np.array_equal(got[:sizes[f]], want[f])
    stride = ptrs[1] - ptrs[0]
    assert stride % 16 == 0
    px = np.empty(raw * n, np.uint8)
    for rnd in range(2):  # the first call parses frame 0 the ordinary way (no header cache yet), the second launches all four at once
        dec.decode_batch(None, device_out=d_out, out_stride=raw, device_in=ptrs[0], in_stride=stride, sizes=sizes)
        assert dec.last_batch() == ((n - 1, 1) if rnd == 0 else (n, 0))
        assert L.gj_hip_memcpy_d2h(px.ctypes.data, d_out, raw * n, None) == 0 and L.gj_hip_stream_sync(None) == 0
        assert all(np.array_equal(px[f * raw:(f + 1) * raw], want_px[f]) for f in range(n))
    # the same streams packed 8 bytes off the 16-byte grid: no batched launch (the marker scan reads whole 16-byte pieces per frame), same pixels
    odd = (max(sizes) + 64 + 15) // 16 * 16 + 8
    d_odd = L.gj_hip_malloc(odd * n + 64)
    for f in range(n):
        assert L.gj_hip_memcpy_d2d(d_odd + f * odd, ptrs[f], sizes[f], None) == 0
    assert L.gj_hip_stream_sync(None) == 0
    dec.decode_batch(None, device_out=d_out, out_stride=raw, device_in=d_odd, in_stride=odd, sizes=sizes)
    assert dec.last_batch() == (0, n)
    assert L.gj_hip_memcpy_d2h(px.ctypes.data, d_out, raw * n, None) == 0 and L.gj_hip_stream_sync(None) == 0
    assert all(np.array_equal(px[f * raw:(f + 1) * raw], want_px[f]) for f in range(n))
    # device streams, host pixels (the binding used to raise) -- the caller says how much room a frame needs
    got_px, _ = dec.decode_batch(None, device_in=ptrs[0], in_stride=stride, sizes=sizes, frame_bytes=raw)
    assert all(np.array_equal(got_px[f], want_px[f]) for f in range(n))
    with pytest.raises(ValueError):
        dec.decode_batch(None, device_in=ptrs[0], in_stride=stride, sizes=sizes)
    # the SAME decoder on another sequence: a smaller image, whose slots are smaller than the cached header's frame -- the call used
    # to fail with "Output stride ... smaller than a decoded frame" --, then one of the old dimensions with another quality (the cached
    # header fits the slots but is not these streams'): frame 0 goes the ordinary way and replaces the cache, the others are batched
    for (w2, h2, q2) in ((320, 240, 75), (640, 480, 50), (640, 480, 75)):
        case2 = ("d2", w2, h2, 1, 1, q2, -1, 0, None, 3)
        p2, pi2 = api_params(gpu_lib, G, case2)
        raw2 = w2 * h2 * 3
        frames2 = np.stack([natural_image(w2, h2, 3, seed=90 + f) for f in range(n)])
        want2 = [O.encode(oracle_image(O, case2), frames2[f]) for f in range(n)]
        assert L.gj_hip_memcpy_h2d(d_frames, frames2.ctypes.data, raw2 * n, None) == 0 and L.gj_hip_stream_sync(None) == 0
        ptrs2, sizes2 = enc.encode_batch_noclone(p2, pi2, d_frames, n, stride=raw2, gpu=True)
        assert sizes2 == [s.size for s in want2]
        dec.decode_batch(None, device_out=d_out, out_stride=raw2, device_in=ptrs2[0], in_stride=ptrs2[1] - ptrs2[0], sizes=sizes2)
        assert dec.last_batch() == (n - 1, 1)
        assert L.gj_hip_memcpy_d2h(px.ctypes.data, d_out, raw2 * n, None) == 0 and L.gj_hip_stream_sync(None) == 0
        assert all(np.array_equal(px[f * raw2:(f + 1) * raw2], O.decode(want2[f])[0]) for f in range(n))
    enc.close()
    dec.close()
    for q in (d_frames, d_out, d_odd):
        L.gj_hip_free(q)


def test_frame_batch_separate_buffers(O, G, gpu_lib):
    """the batch calls for frames, streams and destinations that are separate buffers (gpujpeg_amd_*_batch_ptrs): gathered into / scattered from a
    staging buffer, or coded in place when the buffers happen to lie a constant distance apart; results as always the oracle's"""
    w, h, n = 640, 480, 5
    case = ("p", w, h, 1, 1, 75, -1, 0, None, 3)
    p, pi = api_params(gpu_lib, G, case)
    raw = w * h * 3
    frames = [natural_image(w, h, 3, seed=90 + f) for f in range(n)]
    want = [O.encode(oracle_image(O, case), f) for f in frames]
    want_px = [O.decode(s)[0] for s in want]
    enc, dec = G.Encoder(gpu_lib), G.Decoder(gpu_lib)
    got = enc.encode_batch_ptrs(p, pi, frames)  # five separate numpy buffers
    assert enc.last_batch() == (n, 0) and all(np.array_equal(a, b) for a, b in zip(got, want))
    block = np.concatenate(frames)  # ... and five views a frame apart: coded where they lie
    got = enc.encode_batch_ptrs(p, pi, [block[f * raw:(f + 1) * raw] for f in range(n)])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    for rnd in range(2):
        px, info = dec.decode_batch_ptrs(got, raw + 256)
        assert (info.width, info.height) == (w, h) and all(np.array_equal(a, b) for a, b in zip(px, want_px))
    assert dec.last_batch()[0] >= n - 1
    # streams and destinations a constant distance apart (one block each): used where they are
    stride = (max(s.size for s in got) + 79) & ~15
    sblock, oblock = np.zeros(stride * n, np.uint8), np.zeros((raw + 64) * n, np.uint8)
    for f, s_ in enumerate(got):
        sblock[f * stride:f * stride + s_.size] = s_
    L = gpu_lib.L
    sp = (C.c_void_p * n)(*[sblock.ctypes.data + f * stride for f in range(n)])
    op = (C.c_void_p * n)(*[oblock.ctypes.data + f * (raw + 64) for f in range(n)])
    csz = (C.c_size_t * n)(*[s_.size for s_ in got])
    assert L.gpujpeg_amd_decoder_decode_batch_ptrs(dec.h, sp, csz, n, op, raw, None) == 0
    assert all(np.array_equal(oblock[f * (raw + 64):f * (raw + 64) + raw], want_px[f]) for f in range(n))
    enc.close()
    dec.close()
