"""-m gpu: API behaviour of the drop-in library (ownership, device pointers, statistics, errors)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import api_params, natural_image, oracle_image, psnr

pytestmark = pytest.mark.gpu


def test_device_pointer_inputs_and_outputs(O, G, gpu_lib):
    """A device pointer passed as ENCODER_INPUT_IMAGE is detected and used in place (test/unit/run_tests.c:40-78);
    GPU_IMAGE input, CUSTOM_CUDA_BUFFER output and a device-resident JPEG for the decoder."""
    import torch
    w, h = 640, 360
    raw = natural_image(w, h)
    case = ("x", w, h, 1, 1, 75, -1, 0, None, 3)
    want = O.encode(oracle_image(O, case), raw)
    p, pi = api_params(gpu_lib, G, case)
    d_raw = torch.from_numpy(raw).cuda()
    enc = G.Encoder(gpu_lib)
    assert np.array_equal(enc.encode(p, pi, d_raw.data_ptr(), gpu=True), want)
    inp = G.EncoderInput()
    inp.type, inp.image = G.ENCODER_INPUT_IMAGE, d_raw.data_ptr()  # device pointer disguised as host image
    out, size = C.POINTER(C.c_uint8)(), C.c_size_t()
    assert gpu_lib.L.gpujpeg_encoder_encode(enc.h, C.byref(p), C.byref(pi), C.byref(inp), C.byref(out), C.byref(size)) == 0
    assert np.array_equal(np.ctypeslib.as_array(out, shape=(size.value,)), want)
    # device output of the encoder feeds the decoder directly
    assert enc.set_option("enc_opt_out", "enc_out_val_device") == 0
    jptr, jsize = enc.encode_noclone(p, pi, d_raw.data_ptr(), gpu=True)
    assert jsize == want.size
    d_out = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
    o = G.DecoderOutput()
    o.type, o.data = G.DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, d_out.data_ptr()
    dec = G.Decoder(gpu_lib)
    assert gpu_lib.L.gpujpeg_decoder_decode(dec.h, C.cast(jptr, C.c_void_p), jsize, C.byref(o)) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), O.decode(want)[0])
    assert o.data_size == w * h * 3 and o.param_image.width == w


def test_output_buffer_ownership_and_pinned_option(O, G, gpu_lib):
    w, h = 128, 64
    p, pi = api_params(gpu_lib, G, ("x", w, h, 1, 1, 75, 4, 0, None, 3))
    enc = G.Encoder(gpu_lib)
    a = O.noise(w * h * 3, seed=1)
    b = O.noise(w * h * 3, seed=2)
    pa, na = enc.encode_noclone(p, pi, a)
    first = np.ctypeslib.as_array(pa, shape=(na,)).copy()
    pb, nb = enc.encode_noclone(p, pi, b)
    assert C.addressof(pa.contents) == C.addressof(pb.contents), "the encoder owns one output buffer that is reused"
    assert enc.set_option("enc_opt_out", "enc_out_val_pinned") == 0
    assert np.array_equal(enc.encode(p, pi, a), first)
    assert enc.set_option("enc_opt_out", "bogus") != 0
    assert enc.set_option("no_such_option", "1") != 0


def test_stats_and_kernel_times(O, G, gpu_lib):
    w, h = 1920, 1080
    raw = natural_image(w, h)
    p, pi = api_params(gpu_lib, G, ("x", w, h, 1, 1, 75, -1, 0, None, 3))
    p.perf_stats = 1
    enc = G.Encoder(gpu_lib)
    jpeg = enc.encode(p, pi, raw)
    s = enc.stats()
    assert s.duration_in_gpu > 0 and s.duration_dct_quantization > 0 and s.duration_huffman_coder > 0 and s.duration_memory_to > 0
    kt = enc.kernel_times()
    assert kt is not None and abs(sum(kt) - s.duration_in_gpu) < 0.05
    dec = G.Decoder(gpu_lib)
    dec.init(p, gpu_lib.default_image_parameters())
    dec.decode(jpeg)
    assert dec.stats().duration_in_gpu > 0


def test_error_paths(G, gpu_lib):
    dec = G.Decoder(gpu_lib)
    bad = np.frombuffer(b"this is not a jpeg", np.uint8).copy()
    with pytest.raises(RuntimeError):
        dec.decode(bad)
    img, size = C.POINTER(C.c_uint8)(), C.c_size_t(0)
    gpu_lib.L.gpujpeg_image_load_from_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    assert gpu_lib.L.gpujpeg_image_load_from_file(b"/nonexistent/file.rgb", C.byref(img), C.byref(size)) != 0


def test_cli_round_trip(O, G, gpu_lib, tmp_path):
    """gpujpegtool: synthetic .tst input -> JPEG -> PNM, same files as the reference CLI would produce."""
    import os
    import subprocess
    tool = os.path.join(os.path.dirname(G.PRODUCT_LIB), "gpujpegtool")
    jpg, pnm = tmp_path / "o.jpg", tmp_path / "o.pnm"
    subprocess.check_call([tool, "-q", "80", "-r", "16", "640x360.random.tst", str(jpg)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    raw = O.noise(640 * 360 * 3, seed=12345)
    want = O.encode(O.make_image(640, 360, quality=80, restart_interval=16), raw)
    assert np.array_equal(np.fromfile(jpg, np.uint8), want)
    subprocess.check_call([tool, str(jpg), str(pnm)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    data = open(pnm, "rb").read()
    assert data.startswith(b"P6\n640 360\n255\n")
    assert np.array_equal(np.frombuffer(data[len(b"P6\n640 360\n255\n"):], np.uint8), O.decode(want)[0])


def test_cli_png_in_png_out(O, G, gpu_lib, tmp_path):
    """gpujpegtool with PNG on both ends (the reference goes through stb there): the JPEG equals the oracle's encoding of the
    PNG's pixels, the decoded PNG holds the oracle's decoded pixels."""
    import os
    import subprocess
    Image = pytest.importorskip("PIL.Image")
    tool = os.path.join(os.path.dirname(G.PRODUCT_LIB), "gpujpegtool")
    w, h = 322, 200
    raw = natural_image(w, h, 3, seed=9)
    src, jpg, out = tmp_path / "in.png", tmp_path / "o.jpg", tmp_path / "o.png"
    Image.fromarray(raw.reshape(h, w, 3), "RGB").save(src)
    subprocess.check_call([tool, "-q", "85", str(src), str(jpg)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    want = O.encode(O.make_image(w, h, quality=85), raw)
    assert np.array_equal(np.fromfile(jpg, np.uint8), want)
    subprocess.check_call([tool, str(jpg), str(out)], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    with Image.open(out) as im:
        assert np.array_equal(np.asarray(im).reshape(-1), O.decode(want)[0])


def test_decoder_header_cache(O, G, gpu_lib, monkeypatch):
    """Streams that start with the same header take the speculative path (kernels first, validation after); a stream
    with another header, another output format or a damaged scan structure must fall back and still decode right."""
    import torch
    mk = lambda w, h, q, seed, ri=-1: O.encode(oracle_image(O, ("c", w, h, 1, 1, q, ri, 0, None, 3)), natural_image(w, h, 3, seed=seed))
    a1, a2, a3 = mk(320, 240, 75, 1), mk(320, 240, 75, 2), mk(320, 240, 75, 3)   # same header, different content
    b = mk(320, 240, 60, 4)                                                         # other quantisation tables
    c = mk(336, 240, 75, 5)                                                         # other geometry
    dec = G.Decoder(gpu_lib)
    for jpeg in (a1, a2, a3, b, a1, c, c, a2, b, b):
        px, _ = dec.decode(jpeg)
        assert np.array_equal(px, O.decode(jpeg)[0])
    # device-resident streams: the header is compared on the device
    for jpeg in (a1, a2, b, a3, a3):
        dj = torch.from_numpy(jpeg).cuda()
        out = torch.empty(320 * 240 * 3, dtype=torch.uint8, device="cuda")
        o = G.DecoderOutput()
        o.type, o.data = G.DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, out.data_ptr()
        assert gpu_lib.L.gpujpeg_decoder_decode(dec.h, C.c_void_p(dj.data_ptr()), jpeg.size, C.byref(o)) == 0
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), O.decode(jpeg)[0])
    # changing the requested output format invalidates the cached parse
    dec.set_output_format(G.NONE, G.PIXFMT_NATIVE)
    px, info = dec.decode(a1)
    assert np.array_equal(px, O.decode(a1, info.pixel_format, info.color_space)[0])
    # same header, truncated scan data (no EOI): speculative run, validation fails, careful path reports it like before
    px_full, _ = dec.decode(a2)
    cut = a2[: a2.size - 40].copy()
    try:
        px_cut, _ = dec.decode(cut)
        assert px_cut.size == px_full.size
    except Exception:
        pass  # an error return is acceptable for a damaged stream; it must not crash or hang
    px, _ = dec.decode(a3)
    assert np.array_equal(px, O.decode(a3, info.pixel_format, info.color_space)[0])
    dec.close()


def test_developer_settings_reach_the_coders_that_follow(O, G, gpu_lib):
    """gpujpeg_amd_tuning (round 6): a setting is process-wide and is taken by the coders created AFTER it; coders that exist keep theirs; NULL restores the
    defaults. Observed through the decoder's path counters: GJ_DEC_NO_SPEC keeps a decoder off the speculative path."""
    mk = lambda seed: O.encode(oracle_image(O, ("c", 320, 240, 1, 1, 75, -1, 0, None, 3)), natural_image(320, 240, 3, seed=seed))
    frames = [mk(s) for s in (1, 2, 3, 4)]
    G._settings_from_environment = False  # (tests/conftest.py hands the ENVIRONMENT's settings over before every coder it creates: not here)
    try:
        assert gpu_lib.tuning(None)
        before = G.Decoder(gpu_lib)            # created with the defaults
        assert gpu_lib.tuning("GJ_DEC_NO_SPEC")
        after = G.Decoder(gpu_lib)             # created with the setting
        for jpeg in frames:
            want = O.decode(jpeg)[0]
            assert np.array_equal(before.decode(jpeg)[0], want) and np.array_equal(after.decode(jpeg)[0], want)
        assert before.path_counters()[0] == len(frames) - 1, before.path_counters()  # every frame but the first launched on the cached header
        assert after.path_counters()[0] == 0, after.path_counters()
        assert gpu_lib.tuning(None)
        again = G.Decoder(gpu_lib)
        for jpeg in frames:
            again.decode(jpeg)
        assert again.path_counters()[0] == len(frames) - 1
        for d in (before, after, again):
            d.close()
    finally:
        gpu_lib.tuning(None)
        G._settings_from_environment = True


@pytest.mark.parametrize("pf,comps,mapping", [(1, 3, "210"), (1, 3, "F0Z"), (6, 4, "1230"), (2, 3, "201"), (0, 1, "0")])
def test_channel_remap_and_flip(O, G, gpu_lib, pf, comps, mapping):
    """enc_opt_channel_remap / enc_opt_flipped and their decoder counterparts (src/gpujpeg_preprocessor.cu:455-586,
    src/gpujpeg_postprocessor.cu:445-496): the raw image is permuted in place, the padded planes are flipped."""
    w, h = 152, 100  # height not a multiple of 8: the padding rows take part in the flip, as in the reference
    cs = 1 if pf in (1, 6) else 3
    raw = O.noise(O.raw_size(w, h, pf), seed=7 * pf + comps)
    case = ("x", w, h, pf, cs, 85, 5, 1 if pf == 6 else 0, [(1, 1)] * 4 if pf == 6 else None, 3)
    img = oracle_image(O, case)
    p, pi = api_params(gpu_lib, G, case)
    for flip in (False, True):
        planes = O.preprocess(img, O.channel_remap(img, raw, mapping))
        if flip:
            planes = O.flip_planes(img, planes)
        want = O.encode_from_coefs(img, O.fdct_quant(img, planes))
        enc = G.Encoder(gpu_lib)
        assert enc.set_option("enc_opt_channel_remap", mapping) == 0
        assert enc.set_option("enc_opt_flipped", "1" if flip else "0") == 0
        assert np.array_equal(enc.encode(p, pi, raw), want), ("encode", flip)
        enc.close()
        # decoder: planes flipped before the colour stage, finished image permuted
        s = O.parse(want)
        dplanes = O.idct(s, O.huffman_decode(s, want))
        if flip:
            dplanes = O.flip_planes(s.img, dplanes)
        want_px = O.channel_remap(s.img, O.postprocess(s.img, dplanes), mapping)
        O.lib().gjo_stream_free(C.byref(s))
        dec = G.Decoder(gpu_lib)
        assert gpu_lib.L.gpujpeg_decoder_set_option(dec.h, b"dec_opt_channel_remap", mapping.encode()) == 0
        assert gpu_lib.L.gpujpeg_decoder_set_option(dec.h, b"dec_opt_flipped", b"1" if flip else b"0") == 0
        px, _ = dec.decode(want)
        assert np.array_equal(px, want_px), ("decode", flip)
        dec.close()
    enc = G.Encoder(gpu_lib)
    assert enc.set_option("enc_opt_channel_remap", "0123"[: comps + 1] if comps < 4 else "012") == 0  # wrong channel count: rejected at encode time
    with pytest.raises(Exception):
        enc.encode(p, pi, raw)
    assert enc.set_option("enc_opt_channel_remap", "9") != 0
    enc.close()


def test_encoder_metadata_orientation(O, G, gpu_lib):
    """enc_metadata=orientation=270- switches the default header to SPIFF and the decoder side reports it
    (src/gpujpeg_encoder.c:700-732, src/gpujpeg_writer.c:229-235,456-461, src/gpujpeg_reader.c:449-556)."""
    w, h = 96, 64
    raw = natural_image(w, h)
    case = ("m", w, h, 1, 1, 75, -1, 0, None, 3)
    p, pi = api_params(gpu_lib, G, case)
    enc = G.Encoder(gpu_lib)
    assert enc.set_option("enc_metadata", "orientation=45") != 0
    assert enc.set_option("enc_metadata", "orientation=270-") == 0
    jpeg = enc.encode(p, pi, raw)
    assert bytes(jpeg[2:4]) == b"\xff\xe8" and b"SPIFF" in bytes(jpeg[:32])
    info = G.ImageInfo()
    assert gpu_lib.L.gpujpeg_decoder_get_image_info2(jpeg.ctypes.data_as(C.POINTER(C.c_uint8)), jpeg.size, C.byref(info), -1, 0) == 0
    md = bytes(info.metadata)
    assert md[4] & 1 == 1 and md[0] & 3 == 3 and (md[0] >> 2) & 1 == 1
    px, _ = G.Decoder(gpu_lib).decode(jpeg)  # pixels are not rotated: the orientation is metadata only
    plain = G.Encoder(gpu_lib).encode(p, pi, raw)
    assert np.array_equal(px, G.Decoder(gpu_lib).decode(plain)[0])
    enc.close()


def test_reference_example_programs_run_unchanged(O, G, gpu_lib, tmp_path):
    """The reference's examples/encode_minimal.c and decode_minimal.c, compiled without any change against our public headers and
    linked with our library (oracle/Makefile, binaries under oracle/_ref/examples): the files they write are the oracle's bytes."""
    import os
    import subprocess
    exdir = os.path.join(os.path.dirname(O.REF_PATH), "examples")
    if not os.path.exists(os.path.join(exdir, "encode_minimal")):
        pytest.skip("example binaries were not built (need /root/reference at build time)")
    subprocess.check_call([os.path.join(exdir, "encode_minimal")], cwd=tmp_path, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    w, h = 640, 480
    want = O.encode(O.make_image(w, h, pixel_format=0, color_space=3, restart_interval=8), np.zeros(w * h, np.uint8))  # gpujpeg_set_default_parameters: interval 8
    assert np.array_equal(np.fromfile(tmp_path / "out.jpg", np.uint8), want)
    subprocess.check_call([os.path.join(exdir, "decode_minimal"), "out.jpg"], cwd=tmp_path, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    data = (tmp_path / "out.pnm").read_bytes()
    head = b"P5\n640 480\n255\n"
    assert data.startswith(head)
    assert np.array_equal(np.frombuffer(data[len(head):], np.uint8), O.decode(want)[0])


def test_one_encoder_per_thread(O, G, gpu_lib):
    """test/misc/mt_encode.c: several threads, each with its own stream and encoder (and here a decoder), code HD frames at the same
    time; every result equals the oracle's."""
    import threading
    import torch
    w, h, threads, iterations = 1920, 1080, 4, 6
    raw = (np.arange(w * h * 3, dtype=np.int64) % 255).astype(np.uint8)  # the pattern of the reference's test
    case = ("mt", w, h, 1, 1, 75, 8, 0, None, 3)  # gpujpeg_set_default_parameters: restart interval 8
    want = O.encode(oracle_image(O, case), raw)
    want_px = O.decode(want)[0]
    errors = []

    def worker(t):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            enc, dec = G.Encoder(gpu_lib, stream.cuda_stream), G.Decoder(gpu_lib, stream.cuda_stream)
            p, pi = api_params(gpu_lib, G, case)
            for i in range(iterations):
                jpeg = enc.encode(p, pi, raw)
                if not np.array_equal(jpeg, want):
                    errors.append((t, i, "encode"))
                if not np.array_equal(dec.decode(jpeg)[0], want_px):
                    errors.append((t, i, "decode"))
            enc.close()
            dec.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]


def test_stats_valid_across_host_and_device_inputs(O, G, gpu_lib):
    """test/unit/test_gh_95.c: with perf_stats on, gpujpeg_encoder_get_stats must succeed after every call while host and
    device input buffers alternate (the copy-in timers of the previous call must not be read uninitialised)."""
    import torch

    w, h = 640, 480
    p, pi = api_params(gpu_lib, G, ("s", w, h, 0, 3, 75, -1, 0, None, 3))
    p.perf_stats = 1
    enc = G.Encoder(gpu_lib)
    host = np.zeros(w * h, np.uint8)
    dev = torch.zeros(w * h, dtype=torch.uint8, device="cuda")
    first = None
    for kind in ("cpu", "gpu", "cpu", "gpu", "cpu"):
        if kind == "cpu":
            jpeg = enc.encode(p, pi, host)
        else:
            jptr, jsize = enc.encode_noclone(p, pi, dev.data_ptr(), gpu=True)
            jpeg = np.ctypeslib.as_array(C.cast(jptr, C.POINTER(C.c_uint8)), shape=(jsize,)).copy()
        st = G.DurationStats()
        assert gpu_lib.L.gpujpeg_encoder_get_stats(enc.h, C.byref(st)) == 0, kind
        assert st.duration_in_gpu > 0 and st.duration_huffman_coder >= 0 and st.duration_memory_to >= 0
        first = jpeg if first is None else first
        assert np.array_equal(jpeg, first)
    enc.close()


def test_encoder_custom_exif_tags(O, G, gpu_lib):
    """enc_exif_tag selects the Exif header and adds / replaces tags (src/gpujpeg_encoder.c:773-776, src/gpujpeg_exif.c); the
    entropy-coded data is unaffected, the Exif orientation is read back by the decoder side."""
    w, h = 96, 64
    raw = natural_image(w, h)
    case = ("x", w, h, 1, 1, 75, -1, 0, None, 3)
    p, pi = api_params(gpu_lib, G, case)
    plain = G.Encoder(gpu_lib).encode(p, pi, raw)
    enc = G.Encoder(gpu_lib)
    assert enc.set_option("enc_exif_tag", "NoSuchName=1") != 0
    assert enc.set_option("enc_exif_tag", "0x010F:ASCII=MI355X") == 0
    assert enc.set_option("enc_exif_tag", "Orientation=6") == 0
    assert enc.set_option("enc_exif_tag", "0x829A:RATIONAL=1/250") == 0
    jpeg = enc.encode(p, pi, raw)
    b = bytes(jpeg)
    assert b[2:4] == b"\xff\xe1" and b[6:10] == b"Exif" and b"MI355X\0" in b[:400]
    n = 4 + int.from_bytes(b[4:6], "big")
    assert b[n:] == bytes(plain)[20:], "everything after the application segment equals the JFIF file (APP0 is 18 bytes)"
    info = G.ImageInfo()
    assert gpu_lib.L.gpujpeg_decoder_get_image_info2(jpeg.ctypes.data_as(C.POINTER(C.c_uint8)), jpeg.size, C.byref(info), -1, 0) == 0
    md = bytes(info.metadata)
    assert md[4] & 1 == 1 and md[0] & 3 == 1 and (md[0] >> 2) & 1 == 0  # Exif orientation 6 = a quarter turn, no flip
    assert np.array_equal(G.Decoder(gpu_lib).decode(jpeg)[0], G.Decoder(gpu_lib).decode(plain)[0])
    enc.close()


def test_hostile_tables_and_index(O, G, gpu_lib):
    """Advisor findings of round 1 through the whole decoder: an over-subscribed DHT must be an error return (it used to overrun the
    pinned table buffer), an APP13 index whose offsets are sorted but unrelated to the data must leave the decoder alive, an index
    that fails validation is ignored and the image decodes normally."""
    case = ("h", 320, 240, 1, 1, 80, 4, 0, None, 3)
    jpeg = O.encode(oracle_image(O, case), natural_image(320, 240, 3, seed=9))
    want, _ = O.decode(jpeg)
    dec = G.Decoder(gpu_lib)
    bad = jpeg.copy()
    d = int(np.nonzero((bad[:-1] == 0xFF) & (bad[1:] == 0xC4))[0][0])
    bad[d + 5:d + 21] = np.array([255] + [0] * 15, np.uint8)
    with pytest.raises(Exception):
        dec.decode(bad)
    sos = int(np.nonzero((jpeg[:-1] == 0xFF) & (jpeg[1:] == 0xDA))[0][0])
    rng = np.random.default_rng(1)
    for trial in range(6):
        n = int(rng.integers(2, 200))
        if trial < 3:   # passes validation (non-decreasing, inside the buffer) but points anywhere
            offs = np.sort(rng.integers(0, jpeg.size - sos - 20, size=n))
            offs = np.cumsum(np.maximum(np.diff(np.concatenate([[0], offs])), 2))
            offs = offs[offs < jpeg.size - sos - 20]
        else:           # fails validation -> ignored
            offs = rng.integers(0, 1 << 31, size=n)
        body = offs.astype(">u4").view(np.uint8)
        app13 = np.concatenate([np.array([0xFF, 0xED, (3 + body.size) >> 8, (3 + body.size) & 255, 0], np.uint8), body])
        hostile = np.concatenate([jpeg[:sos], app13, jpeg[sos:]])
        try:
            px, _ = dec.decode(hostile)
            if trial >= 3:
                assert np.array_equal(px, want), "a rejected index must not change the result"
        except Exception:
            assert trial < 3
        assert np.array_equal(dec.decode(jpeg)[0], want)


def test_damaged_streams_do_not_crash(O, G, gpu_lib):
    """Corrupted entropy data, truncated files and garbage after the headers: the decoder may fail or return garbage pixels, but it
    must return (no out-of-bounds access, no endless loop in the synchronisation rounds) and keep working afterwards."""
    rng = np.random.default_rng(5)
    good = {}
    for name, w, h, ri, il, ss in [("a", 320, 240, -1, 0, None), ("b", 256, 192, 0, 0, None), ("c", 320, 240, 3, 1, [(2, 2), (1, 1), (1, 1)])]:
        case = (name, w, h, 1, 1, 80, ri, il, ss, 3)
        good[name] = O.encode(oracle_image(O, case), natural_image(w, h, 3, seed=w))
    dec = G.Decoder(gpu_lib)
    for name, jpeg in good.items():
        hdr = 700  # leave the headers alone: header damage is the host parser's business (tests/test_host_logic.py)
        for trial in range(12):
            bad = jpeg.copy()
            if trial >= 6:  # many more stray / damaged markers: 0xFF bytes anywhere
                idx = rng.integers(hdr, bad.size - 2, size=4 * (trial - 5))
                bad[idx] = 0xFF
            elif trial < 3:  # random bytes in the entropy-coded part (0xFF avoided: markers would change the segment structure)
                idx = rng.integers(hdr, bad.size - 2, size=40)
                bad[idx] = rng.integers(0, 255, size=idx.size, dtype=np.uint8)
            elif trial == 3:  # stray restart markers
                idx = rng.integers(hdr, bad.size - 4, size=5)
                for i in idx:
                    bad[i], bad[i + 1] = 0xFF, 0xD0 + int(rng.integers(0, 8))
            elif trial == 4:  # truncated
                bad = bad[: bad.size // 2].copy()
            else:  # zeros after the headers
                bad[hdr + 50:] = 0
            try:
                dec.decode(bad)
            except Exception:
                pass
        px, _ = dec.decode(jpeg)  # the decoder object is still usable and correct
        assert np.array_equal(px, O.decode(jpeg)[0]), name
    dec.close()


@pytest.mark.parametrize("ext", ["bmp", "tga", "png"])
@pytest.mark.parametrize("pf,comps", [(1, 3), (6, 4), (0, 1)])
def test_bmp_tga_roundtrip(gpu_lib, G, tmp_path, ext, pf, comps):
    lib = gpu_lib
    """gpujpeg_image_save_to_file / _get_properties / _load_from_file for BMP and TGA (the reference delegates these to stb,
    src/utils/image_delegate.c:188-330): 1, 3 and 4 channels survive a round trip; hand-made RLE TGA and bottom-up BMP read back."""
    import ctypes as C
    w, h = 37, 21  # odd width: BMP rows are padded to 4 bytes
    rng = np.random.default_rng(comps)
    img = rng.integers(0, 256, size=w * h * comps, dtype=np.uint8)
    pi = lib.default_image_parameters()
    pi.width, pi.height, pi.pixel_format, pi.color_space = w, h, pf, 1 if comps > 1 else 3
    path = str(tmp_path / f"x.{ext}").encode()
    assert lib.L.gpujpeg_image_save_to_file(path, img.ctypes.data_as(C.POINTER(C.c_uint8)), img.size, C.byref(pi)) == 0
    got = lib.default_image_parameters()
    assert lib.L.gpujpeg_image_get_properties(path, C.byref(got), 1) == 0
    want_pf = pf if not (ext == "bmp" and comps == 1) else 1  # grey is written as 24-bit BMP (like stb_image_write does)
    assert (got.width, got.height, got.pixel_format) == (w, h, want_pf)
    data, size = C.POINTER(C.c_uint8)(), C.c_size_t(0)
    assert lib.L.gpujpeg_image_load_from_file(path, C.byref(data), C.byref(size)) == 0
    back = np.ctypeslib.as_array(data, shape=(size.value,)).copy()
    lib.L.gpujpeg_image_destroy(data)
    if ext == "bmp" and comps == 1:
        assert np.array_equal(back.reshape(-1, 3)[:, 0], img) and np.array_equal(back.reshape(-1, 3)[:, 2], img)
    else:
        assert np.array_equal(back, img)


def test_tga_rle_and_origin(gpu_lib, G, tmp_path):
    lib = gpu_lib
    import ctypes as C
    w, h = 5, 3
    px = np.arange(w * h * 3, dtype=np.uint8).reshape(h, w, 3)  # RGB, top-down
    # run-length coded (type 10), bottom-left origin: rows bottom-up, one raw packet per row followed by nothing
    body = b""
    for y in range(h - 1, -1, -1):
        body += bytes([w - 1]) + px[y][:, ::-1].tobytes()  # raw packet of w pixels, BGR
    hdr = bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, w, 0, h, 0, 24, 0])
    path = tmp_path / "r.tga"
    path.write_bytes(hdr + body)
    data, size = C.POINTER(C.c_uint8)(), C.c_size_t(0)
    assert lib.L.gpujpeg_image_load_from_file(str(path).encode(), C.byref(data), C.byref(size)) == 0
    back = np.ctypeslib.as_array(data, shape=(size.value,)).copy()
    lib.L.gpujpeg_image_destroy(data)
    assert np.array_equal(back, px.reshape(-1))
    # repeat packets: a constant image in a few bytes
    path.write_bytes(bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, w, 0, h, 0, 24, 0x20]) + bytes([0x80 | (w * h - 1), 3, 2, 1]))
    assert lib.L.gpujpeg_image_load_from_file(str(path).encode(), C.byref(data), C.byref(size)) == 0
    back = np.ctypeslib.as_array(data, shape=(size.value,)).copy()
    lib.L.gpujpeg_image_destroy(data)
    assert np.array_equal(back, np.tile(np.array([1, 2, 3], np.uint8), w * h))


@pytest.mark.parametrize("mode,subsampling", [("L", None), ("RGB", 0), ("RGB", 1), ("RGB", 2)])
def test_foreign_jpegs_from_libjpeg(O, G, gpu_lib, tmp_path, mode, subsampling):
    """Baseline files written by another encoder (libjpeg-turbo through PIL): standard and optimised Huffman tables, no restart
    markers (one long segment per scan, decoded piece by piece) and restart markers every MCU row, interleaved 4:2:0 / 4:2:2 /
    4:4:4 and grey. Our pixels equal the oracle's (the reference reader + decoder) and are close to libjpeg's own decoding."""
    Image = pytest.importorskip("PIL.Image")
    import io
    w, h = 487, 331
    img = natural_image(w, h, 3, seed=21).reshape(h, w, 3)
    im = Image.fromarray(img, "RGB").convert(mode)
    dec = G.Decoder(gpu_lib)
    for quality in (50, 92):
        for optimize in (False, True):
            for restart_rows in (0, 1, 3):
                kw = {"quality": quality, "optimize": optimize}
                if subsampling is not None:
                    kw["subsampling"] = subsampling
                if restart_rows:
                    kw["restart_marker_rows"] = restart_rows
                buf = io.BytesIO()
                try:
                    im.save(buf, "JPEG", **kw)
                except TypeError:
                    pytest.skip("this PIL cannot write restart markers")
                jpeg = np.frombuffer(buf.getvalue(), np.uint8).copy()
                px, info = dec.decode(jpeg)
                want, winfo = O.decode(jpeg)
                assert (info.width, info.height) == (w, h)
                assert np.array_equal(px, want), (mode, subsampling, quality, optimize, restart_rows)
                theirs = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB" if mode == "RGB" else "L")).reshape(-1)
                assert psnr(px, theirs) > 30.0, (mode, subsampling, quality, optimize, restart_rows)
    dec.close()


@pytest.mark.gpu
def test_host_buffers_after_device_reset(tmp_path):
    """gpujpeg_device_reset destroys the device's streams, the process's copy lanes (gj_runtime.hip) among them: coders created afterwards get
    new ones. In a process of its own -- the reset would take this session's torch context with it."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from gpujpeg_amd import libgpujpeg as G
lib = G.Library()
assert lib.L.gpujpeg_init_device(0, 0) == 0
w, h = 1024, 768  # 2.4 MB of RGB: above the lanes' threshold
raw = (np.arange(w * h * 3, dtype=np.int64) %% 251).astype(np.uint8)
def roundtrip():
    p, pi = lib.default_parameters(), lib.default_image_parameters()
    pi.width, pi.height = w, h
    enc, dec = G.Encoder(lib), G.Decoder(lib)
    jpeg = enc.encode(p, pi, raw)
    px, _ = dec.decode(jpeg)
    enc.close(); dec.close()
    return jpeg, px
a = roundtrip()
lib.L.gpujpeg_device_reset()
assert lib.L.gpujpeg_init_device(0, 0) == 0
b = roundtrip()
assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
