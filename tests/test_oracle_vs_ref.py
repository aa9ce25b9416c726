"""Pins the CPU restatement (oracle/gj_oracle.c) against the reference's own code.

oracle/_ref/libgpujpeg_ref.so is the reference's host C files (driver, geometry, tables, JFIF writer/reader,
CPU Huffman coders) compiled unmodified from /root/reference against a host-memory CUDA stub; its CUDA-only
stages are provided by the restatement. So for identical parameters and pixels:
  * complete JPEG bytes must match  -> geometry, tables, header bytes, Huffman, stuffing, RSTn, stitching
  * the reference's CPU Huffman coder re-encoding the same coefficients must give the same file
  * decode through reference reader + reference CPU Huffman decoder must match the restated parser/decoder
"""
import ctypes as C

import numpy as np
import pytest

from conftest import CASES, api_params, make_raw, oracle_image


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_encode_bytes_and_decode_pixels(O, G, ref, case):
    raw = make_raw(O, case)
    p, pi = api_params(ref, G, case)
    enc = G.Encoder(ref)
    jpeg = enc.encode(p, pi, raw)
    want = O.encode(oracle_image(O, case), raw)
    assert np.array_equal(jpeg, want)

    # the reference CPU Huffman coder (src/gpujpeg_huffman_cpu_encoder.c:297) on the same coefficients
    out, size = C.POINTER(C.c_uint8)(), C.c_size_t()
    ref.L.gjref_reencode_cpu_huffman.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    assert ref.L.gjref_reencode_cpu_huffman(enc.h, C.byref(out), C.byref(size)) == 0
    cpu = np.ctypeslib.as_array(out, shape=(size.value,)).copy()
    if case[8] is None or True:
        assert np.array_equal(cpu, jpeg), "GPU-path segment stitching differs from the reference CPU Huffman stream"

    # decode: default output format, then the native one
    dec = G.Decoder(ref)
    px, info = dec.decode(jpeg)
    opx, oinfo = O.decode(want)
    assert (info.width, info.height, info.pixel_format, info.color_space) == (oinfo.width, oinfo.height, oinfo.pixel_format, oinfo.color_space)
    assert np.array_equal(px, opx)


@pytest.mark.parametrize("case", [c for c in CASES if c[6] != 0][:8], ids=lambda c: c[0])
def test_segment_info_index(O, G, ref, case):
    """APP13 segment index (src/gpujpeg_writer.c:522-623) written and consumed."""
    raw = make_raw(O, case)
    p, pi = api_params(ref, G, case, segment_info=1)
    jpeg = G.Encoder(ref).encode(p, pi, raw)
    want = O.encode(oracle_image(O, case, segment_info=1), raw)
    assert np.array_equal(jpeg, want)
    px, _ = G.Decoder(ref).decode(jpeg)
    assert np.array_equal(px, O.decode(want)[0])


@pytest.mark.parametrize("quality", [1, 10, 25, 49, 50, 51, 75, 90, 99, 100])
def test_quantisation_tables(O, ref, quality):
    ref.L.gjref_quant_tables.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_uint16)]
    for t in (0, 1):
        raw, fwd, inv = (C.c_uint8 * 64)(), (C.c_float * 64)(), (C.c_uint16 * 64)()
        ref.L.gjref_quant_tables(t, quality, raw, fwd, inv)
        oraw, ofwd, oinv = (C.c_uint8 * 64)(), (C.c_float * 64)(), (C.c_uint16 * 64)()
        O.lib().gjo_quant_table(t, quality, oraw, ofwd, oinv)
        assert bytes(raw) == bytes(oraw)
        assert np.array_equal(np.array(fwd[:], np.float32).view(np.uint32), np.array(ofwd[:], np.float32).view(np.uint32)), "forward table must be bit-identical"
        assert list(inv) == list(oinv)


@pytest.mark.parametrize("pf,w,h", [(1, 641, 481), (3, 322, 77), (5, 33, 35), (4, 33, 35), (2, 10, 10), (0, 99, 3), (6, 17, 9)])
def test_output_format_requests(O, G, ref, pf, w, h):
    """Decoder output formats other than the default: native planar / packed layouts and colour spaces."""
    cs = 1 if pf in (1, 6) else 3
    raw = O.noise(O.raw_size(w, h, pf), seed=pf + w)
    case = ("x", w, h, pf, cs, 80, 4, 1 if pf != 0 else 0, None, 3)
    p, pi = api_params(ref, G, case)
    jpeg = G.Encoder(ref).encode(p, pi, raw)
    for opf, ocs in [(pf, cs), (1, 1), (G.PIXFMT_NATIVE, G.NONE)]:
        dec = G.Decoder(ref)
        dec.set_output_format(ocs, opf)
        px, info = dec.decode(jpeg)
        opx, oinfo = O.decode(jpeg, info.pixel_format, info.color_space)
        assert np.array_equal(px, opx), (opf, ocs)
