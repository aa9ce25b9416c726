"""-m "not gpu": the raster file formats behind gpujpeg_image_save_to_file / _load_from_file / _get_properties, DIFFERENTIALLY against
the reference's own writers and readers (src/utils/image_delegate.c:119-632, src/utils/pam.c, src/utils/y4m.c, src/gpujpeg_common.c:1217-1370 --
compiled where they lie into oracle/_ref/libgpujpeg_ref.so, stb included): for PNM / PGM / PPM / PAM / Y4M / BMP / TGA / raw and the `.tst`
pattern generator both libraries write the same image and must produce equal bytes (PNG: equal decoded pixels -- the reference's deflate is stb's,
ours has fixed codes), report the same properties for each other's files, and read each other's files to the same samples; what the reference
rejects the product rejects.
The product's loader returns pinned memory, which needs a device: loading goes through the CPU execution model of the same sources
(tests/hipemu: gj_image_io.c and gj_image_png.c are plain C, identical in both builds); saving and probing use the product library itself."""
import ctypes as C
import os

import numpy as np
import pytest

from test_emu_parity import emu_lib  # noqa: F401  (fixture: the product's sources on the CPU execution model)

U8, P012, P0P1P2_444, P1020_422, P0P1P2_422, P0P1P2_420, P0123 = 0, 1, 2, 3, 4, 5, 6
RGB, BT601, BT601_256, BT709 = 1, 2, 3, 4


def raw_size(w, h, pf):
    return {U8: w * h, P012: 3 * w * h, P0P1P2_444: 3 * w * h, P1020_422: 2 * ((w + 1) // 2 * 2) * h, P0P1P2_422: w * h + 2 * ((w + 1) // 2) * h,
            P0P1P2_420: w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2), P0123: 4 * w * h}[pf]


def bind(lib):
    L = lib.L
    L.gpujpeg_image_save_to_file.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_void_p]
    L.gpujpeg_image_save_to_file.restype = C.c_int
    L.gpujpeg_image_load_from_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.gpujpeg_image_load_from_file.restype = C.c_int
    L.gpujpeg_image_get_properties.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    L.gpujpeg_image_get_properties.restype = C.c_int
    L.gpujpeg_image_destroy.argtypes = [C.POINTER(C.c_uint8)]
    return L


def save(lib, path, img, w, h, pf, cs):
    L = bind(lib)
    pi = lib.default_image_parameters()
    pi.width, pi.height, pi.pixel_format, pi.color_space = w, h, pf, cs
    return L.gpujpeg_image_save_to_file(str(path).encode(), img.ctypes.data_as(C.POINTER(C.c_uint8)), img.size, C.byref(pi))


def load(lib, path):
    L = bind(lib)
    data, size = C.POINTER(C.c_uint8)(), C.c_size_t(0)
    rc = L.gpujpeg_image_load_from_file(str(path).encode(), C.byref(data), C.byref(size))
    if rc != 0:
        return rc, None
    out = np.ctypeslib.as_array(data, shape=(size.value,)).copy()
    L.gpujpeg_image_destroy(data)
    return 0, out


def probe(lib, path, exists=1):
    L = bind(lib)
    pi = lib.default_image_parameters()
    pi.width, pi.height, pi.pixel_format, pi.color_space = 0, 0, -1, 0  # (what the file does not say stays as it was: compare that too)
    rc = L.gpujpeg_image_get_properties(str(path).encode(), C.byref(pi), exists)
    return rc, (pi.width, pi.height, pi.pixel_format, pi.color_space)


def image(w, h, pf, seed):
    return np.random.default_rng(seed).integers(0, 256, size=raw_size(w, h, pf), dtype=np.uint8)


# (extension, pixel format, colour space): what gpujpegtool can be asked to write -- including combinations the reference refuses
WRITE_MATRIX = [
    ("ppm", P012, RGB), ("pnm", P012, RGB), ("pnm", U8, BT601_256), ("pgm", U8, BT601_256), ("pam", U8, BT601_256), ("pam", P012, RGB), ("pam", P0123, RGB),
    ("pnm", P0123, RGB), ("ppm", U8, BT601_256), ("pgm", P012, RGB), ("pam", P012, BT601_256), ("pam", P0P1P2_444, RGB), ("pnm", P1020_422, BT601_256),
    ("y4m", U8, BT601_256), ("y4m", P0P1P2_420, BT601_256), ("y4m", P0P1P2_422, BT601_256), ("y4m", P0P1P2_444, BT601_256), ("y4m", P0P1P2_420, BT601),
    ("y4m", P0P1P2_444, BT709), ("y4m", P012, RGB), ("y4m", P1020_422, BT601_256),
    ("bmp", P012, RGB), ("bmp", P0123, RGB), ("bmp", U8, BT601_256), ("tga", P012, RGB), ("tga", P0123, RGB), ("tga", U8, BT601_256), ("tga", P0P1P2_444, RGB),
    ("png", P012, RGB), ("png", P0123, RGB), ("png", U8, BT601_256),
    ("rgb", P012, RGB), ("rgba", P0123, RGB), ("yuv", P0P1P2_444, BT601_256), ("uyvy", P1020_422, BT601_256), ("i420", P0P1P2_420, BT601_256), ("r", U8, BT601_256),
    ("raw", P012, RGB),
]


@pytest.mark.parametrize("w,h", [(37, 21), (64, 48), (1, 1), (2, 3)])
@pytest.mark.parametrize("ext,pf,cs", WRITE_MATRIX, ids=[f"{e}-pf{p}-cs{c}" for e, p, c in WRITE_MATRIX])
def test_written_files_equal_the_references(lib, emu_lib, _ref_lib, tmp_path, ext, pf, cs, w, h):  # noqa: F811
    ref = _ref_lib
    img = image(w, h, pf, seed=w * 131 + h + pf)
    ours, theirs = tmp_path / f"ours.{ext}", tmp_path / f"theirs.{ext}"
    rc_ref = save(ref, theirs, img, w, h, pf, cs)
    rc_our = save(lib, ours, img, w, h, pf, cs)
    assert (rc_our == 0) == (rc_ref == 0), f"reference returned {rc_ref}, product {rc_our}"
    if rc_ref != 0:
        return
    a, b = ours.read_bytes(), theirs.read_bytes()
    if ext != "png":
        assert a == b, f"{ext}: {len(a)} B against the reference's {len(b)} B, first difference at {next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))}"
    # each library's view of both files: the same properties, the same samples
    results = []
    for reader, can_load in ((ref, True), (emu_lib, True)):
        for path in (ours, theirs):
            rc_p, props = probe(reader if reader is ref else lib, path)
            rc_l, px = load(reader, path) if can_load else (0, None)
            results.append((rc_p, props, rc_l, px))
    first = results[0]
    for r in results[1:]:
        assert (r[0], r[1], r[2]) == (first[0], first[1], first[2]), (results[0][:3], r[:3])
        if first[3] is not None:
            assert np.array_equal(r[3], first[3])
    # and what was written is what comes back (formats that keep every sample)
    if first[2] == 0 and ext not in ("bmp",) and not (ext == "tga" and pf == U8 and False):
        if first[3].size == img.size:
            assert np.array_equal(first[3], img) or ext in ("bmp", "tga", "png")  # (BMP / TGA / PNG: channel order and alpha rules of the format)


TST_NAMES = [
    "64x32.tst", "37x21.gradient.tst", "64x32.noise.tst", "33x17.random.tst", "33x17.random_7.tst", "16x16.blank.tst", "16x16.blank_200.tst", "16x16.blank_0x40.tst",
    "40x24.c_ycbcr-jpeg.p_422-u8-p1020.tst", "40x24.c_ycbcr-jpeg.p_420-u8-p0p1p2.random.tst", "40x24.p_u8.random.tst", "24x8.p_4444-u8-p0123.random.tst",
    "24x8.c_ycbcr-bt709.p_444-u8-p0p1p2.noise.tst", "1119x561.random.c_rgb.tst", "8x8.c_nonsense.tst", "8x8.p_nonsense.tst", "8x8.wibble.tst", "x8.tst", "8x.tst", "8x0.tst",
]


@pytest.mark.parametrize("name", TST_NAMES)
def test_tst_patterns_equal_the_references(lib, emu_lib, _ref_lib, name):  # noqa: F811
    """The `.tst` pattern generator (src/utils/image_delegate.c:383-632): same properties, same bytes, same refusals."""
    if name in ("x8.tst", "8x.tst"):
        pytest.skip("the reference prints its usage and reads a null pointer on these names")
    rc_r, props_r = probe(_ref_lib, name)
    rc_o, props_o = probe(lib, name)
    assert (rc_r == 0) == (rc_o == 0), (rc_r, rc_o)
    if rc_r == 0:
        assert props_r == props_o
    lr, pr = load(_ref_lib, name)
    lo, po = load(emu_lib, name)
    assert (lr == 0) == (lo == 0), (lr, lo)
    if lr == 0:
        if ".noise" in name:  # the reference seeds this pattern from the clock (image_delegate.c:421): the size is all that can be compared
            assert pr.size == po.size
        else:
            assert np.array_equal(pr, po)


@pytest.mark.parametrize("name", ["a.rgb", "a.rgba", "a.yuv", "a.yuva", "a.uyvy", "a.i420", "a.r", "a.raw", "a.jpg", "a.jpeg", "a.jfif", "a.bmp", "a.gif", "a.png", "a.tga", "a.pnm",
                                  "a.pgm", "a.ppm", "a.pam", "a.y4m", "a.tst", "a.XXX", "a.pbm", "a", "a.", ".rgb", "dir.ppm/a", "A.PPM", "a.Y4M"])
def test_properties_of_a_file_to_be_written(lib, _ref_lib, name):
    """gpujpeg_image_get_properties(file_exists = 0) decides the decoder's output format from the NAME of the file it is going to write
    (src/gpujpeg_common.c:1316-1370, src/main.c): identical answers."""
    if name.endswith(".tst"):
        pytest.skip("a name without dimensions: the reference's parser reads a null pointer")
    rc_r, props_r = probe(_ref_lib, name, exists=0)
    rc_o, props_o = probe(lib, name, exists=0)
    assert rc_r == rc_o and props_r == props_o, (name, rc_r, props_r, rc_o, props_o)
    L, R = bind(lib), bind(_ref_lib)
    L.gpujpeg_image_get_file_format.argtypes = R.gpujpeg_image_get_file_format.argtypes = [C.c_char_p]
    assert L.gpujpeg_image_get_file_format(name.encode()) == R.gpujpeg_image_get_file_format(name.encode())


def test_foreign_files_read_like_the_reference(lib, emu_lib, _ref_lib, tmp_path):  # noqa: F811
    """Files other programs write: PNM with comments and odd white space, ASCII variants, 16-bit samples, PAM with extra header lines and
    tuple types, Y4M with frame parameters / interlacing / colour-range tags, truncated files: the reference's readers (src/utils/pam.c,
    src/utils/y4m.c) decide what is accepted, and with which samples."""
    rng = np.random.default_rng(3)
    px3 = rng.integers(0, 256, 5 * 4 * 3, dtype=np.uint8).tobytes()
    px1 = rng.integers(0, 256, 5 * 4, dtype=np.uint8).tobytes()
    px4 = rng.integers(0, 256, 5 * 4 * 4, dtype=np.uint8).tobytes()
    px16 = rng.integers(0, 256, 5 * 4 * 3 * 2, dtype=np.uint8).tobytes()
    yuv420 = rng.integers(0, 256, 6 * 4 + 2 * 3 * 2, dtype=np.uint8).tobytes()
    files = {
        "plain.ppm": b"P6\n5 4\n255\n" + px3,
        "comment.ppm": b"P6\n# made by hand\n5 4\n# another\n255\n" + px3,
        "spaces.ppm": b"P6 5\t4   255\n" + px3,
        "crlf.ppm": b"P6\r\n5 4\r\n255\r\n" + px3,
        "gray.pgm": b"P5\n5 4\n255\n" + px1,
        "gray_as_pnm.pnm": b"P5\n5 4\n255\n" + px1,
        "ascii.ppm": b"P3\n5 4\n255\n" + b" ".join(str(v).encode() for v in px3) + b"\n",
        "ascii.pgm": b"P2\n5 4\n255\n" + b" ".join(str(v).encode() for v in px1) + b"\n",
        "bitmap.pnm": b"P4\n8 2\n\xAA\x55",
        "deep.ppm": b"P6\n5 4\n65535\n" + px16,
        "maxval100.ppm": b"P6\n5 4\n100\n" + px3,
        "short.ppm": b"P6\n5 4\n255\n" + px3[:-7],
        "nodims.ppm": b"P6\n",
        "rgb.pam": b"P7\nWIDTH 5\nHEIGHT 4\nDEPTH 3\nMAXVAL 255\nTUPLTYPE RGB\nENDHDR\n" + px3,
        "rgba.pam": b"P7\nWIDTH 5\nHEIGHT 4\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n" + px4,
        "gray.pam": b"P7\nWIDTH 5\nHEIGHT 4\nDEPTH 1\nMAXVAL 255\nTUPLTYPE GRAYSCALE\nENDHDR\n" + px1,
        "order.pam": b"P7\nHEIGHT 4\n# c\nDEPTH 3\nWIDTH 5\nMAXVAL 255\nENDHDR\n" + px3,
        "depth2.pam": b"P7\nWIDTH 5\nHEIGHT 4\nDEPTH 2\nMAXVAL 255\nTUPLTYPE GRAYSCALE_ALPHA\nENDHDR\n" + px1 * 2,
        "deep.pam": b"P7\nWIDTH 5\nHEIGHT 4\nDEPTH 3\nMAXVAL 65535\nTUPLTYPE RGB\nENDHDR\n" + px16,
        "noend.pam": b"P7\nWIDTH 5\nHEIGHT 4\nDEPTH 3\nMAXVAL 255\n" + px3,
        "c420.y4m": b"YUV4MPEG2 W6 H4 F25:1 Ip A1:1 C420\nFRAME\n" + yuv420,
        "c420jpeg.y4m": b"YUV4MPEG2 W6 H4 F30000:1001 C420jpeg XYSCSS=420JPEG\nFRAME\n" + yuv420,
        "full.y4m": b"YUV4MPEG2 W6 H4 C420 XCOLORRANGE=FULL\nFRAME\n" + yuv420,
        "limited.y4m": b"YUV4MPEG2 W6 H4 C420 XCOLORRANGE=LIMITED\nFRAME\n" + yuv420,
        "mono.y4m": b"YUV4MPEG2 W5 H4 Cmono\nFRAME\n" + px1,
        "c444.y4m": b"YUV4MPEG2 W5 H4 C444\nFRAME\n" + px3,
        "c422.y4m": b"YUV4MPEG2 W6 H4 C422\nFRAME\n" + rng.integers(0, 256, 6 * 4 * 2, dtype=np.uint8).tobytes(),
        "nocs.y4m": b"YUV4MPEG2 W6 H4\nFRAME\n" + yuv420,
        "alpha.y4m": b"YUV4MPEG2 W5 H4 C444alpha\nFRAME\n" + px4,
        "p10.y4m": b"YUV4MPEG2 W6 H4 C420p10\nFRAME\n" + yuv420 * 2,
        "frameparams.y4m": b"YUV4MPEG2 W6 H4 C420\nFRAME Ip\n" + yuv420,
        "noframe.y4m": b"YUV4MPEG2 W6 H4 C420\n" + yuv420,
        "short.y4m": b"YUV4MPEG2 W6 H4 C420\nFRAME\n" + yuv420[:-5],
        "junk.y4m": b"not a y4m file at all",
    }
    for name, data in files.items():
        path = tmp_path / name
        path.write_bytes(data)
        rc_r, props_r = probe(_ref_lib, path)
        rc_o, props_o = probe(lib, path)
        assert (rc_r == 0) == (rc_o == 0), (name, "properties", rc_r, rc_o)
        if rc_r == 0:
            assert props_r == props_o, (name, props_r, props_o)
        lo, po = load(emu_lib, path)
        if rc_r != 0:  # the reference's loaders assert on what its probe refused (image_delegate.c:151): this library must refuse, not abort
            assert lo != 0, (name, "load of a file whose properties were refused")
            continue
        lr, pr = load(_ref_lib, path)
        assert (lr == 0) == (lo == 0), (name, "load", lr, lo)
        if lr == 0:
            assert np.array_equal(pr, po), name
