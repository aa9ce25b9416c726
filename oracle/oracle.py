"""ctypes wrapper around oracle/libgjoracle.so (the CPU restatement) and, when present,
oracle/_ref/libgpujpeg_ref.so (the reference's own host C code built against a host-memory CUDA stub).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package gpujpeg_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgjoracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libgpujpeg_ref.so")                 # contraction off (CPU, cudaemu)
REF_FMA_PATHS = {"gcc": os.path.join(HERE, "_ref", "libgpujpeg_ref_fma_gcc.so"), "clang": os.path.join(HERE, "_ref", "libgpujpeg_ref_fma_clang.so")}
REFHIP_PATH = os.path.join(HERE, "_ref", "libgpujpeg_refhip.so")           # hipcc gfx950, the fusion-map pin (GPU box)
REFHIP_SLP_PATH = os.path.join(HERE, "_ref", "libgpujpeg_refhip_slp.so")   # hipcc defaults (SLP vectoriser on)

CS_NONE, CS_RGB, CS_BT601, CS_BT601_256, CS_BT709, CS_YUV = range(6)
PF_U8, PF_444_P012, PF_444_P0P1P2, PF_422_P1020, PF_422_P0P1P2, PF_420_P0P1P2, PF_4444_P0123 = range(7)
MAX_COMP = 4


class Comp(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "type", "h", "v", "width", "height", "data_width", "data_height", "mcu_size_x", "mcu_size_y", "mcu_size",
        "mcu_count_x", "mcu_count_y", "mcu_count", "segment_mcu_count", "segment_count")] + [("data_offset", C.c_size_t)]


class Image(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("width_padding", C.c_int), ("pixel_format", C.c_int),
        ("color_space", C.c_int), ("comp_count", C.c_int), ("samp_h", C.c_int * MAX_COMP), ("samp_v", C.c_int * MAX_COMP),
        ("interleaved", C.c_int), ("restart_interval", C.c_int), ("quality", C.c_int), ("color_space_internal", C.c_int),
        ("segment_info", C.c_int), ("header_type", C.c_int),
        ("max_h", C.c_int), ("max_v", C.c_int), ("comp", Comp * MAX_COMP), ("data_size", C.c_size_t),
        ("raw_size", C.c_size_t), ("mcu_count", C.c_int), ("segment_count", C.c_int), ("segment_mcu_count", C.c_int),
        ("block_count", C.c_int), ("scan_count", C.c_int),
    ]


class Stream(C.Structure):
    _fields_ = [
        ("img", Image), ("qraw", (C.c_uint8 * 64) * 4), ("qinv", (C.c_uint16 * 64) * 4), ("qmap", C.c_int * MAX_COMP),
        ("hbits", ((C.c_uint8 * 17) * 2) * 4), ("hvals", ((C.c_uint8 * 256) * 2) * 4), ("hmap", (C.c_int * 2) * MAX_COMP),
        ("comp_id", C.c_uint8 * MAX_COMP), ("seg_count", C.c_int), ("seg_offset", C.POINTER(C.c_size_t)),
        ("seg_size", C.POINTER(C.c_size_t)), ("seg_scan", C.POINTER(C.c_int)), ("seg_index_in_scan", C.POINTER(C.c_int)),
    ]


def build(force=False):
    """Compile the restatement (always) and oracle/_ref (only where /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE] + (["-B"] if force else []))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        u8p, i16p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16)
        L.gjo_raw_size.restype = C.c_size_t
        L.gjo_raw_size.argtypes = [C.c_int] * 4
        L.gjo_image_init.argtypes = [C.POINTER(Image)]
        L.gjo_adjust_encoder_params.argtypes = [C.POINTER(Image)]
        L.gjo_preprocess.argtypes = [C.POINTER(Image), u8p, u8p]
        L.gjo_fdct_quant.argtypes = [C.POINTER(Image), u8p, i16p]
        L.gjo_huffman_encode_segment.restype = C.c_size_t
        L.gjo_huffman_encode_segment.argtypes = [C.POINTER(Image), i16p, C.c_int, u8p]
        L.gjo_write_header.restype = C.c_size_t
        L.gjo_write_header.argtypes = [C.POINTER(Image), u8p]
        L.gjo_encode_from_coefs.restype = C.c_size_t
        L.gjo_encode_from_coefs.argtypes = [C.POINTER(Image), i16p, u8p, C.c_size_t]
        L.gjo_encode.restype = C.c_size_t
        L.gjo_encode.argtypes = [C.POINTER(Image), u8p, u8p, C.c_size_t]
        L.gjo_parse.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.POINTER(Stream)]
        L.gjo_stream_free.argtypes = [C.POINTER(Stream)]
        L.gjo_huffman_decode.argtypes = [C.POINTER(Stream), u8p, i16p]
        L.gjo_idct.argtypes = [C.POINTER(Stream), i16p, u8p]
        L.gjo_postprocess.argtypes = [C.POINTER(Image), u8p, u8p]
        L.gjo_decode.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t, C.POINTER(Image)]
        L.gjo_color_transform.argtypes = [C.c_int, C.c_int, u8p]
        L.gjo_flip_planes.argtypes = [C.POINTER(Image), u8p]
        L.gjo_parse_channel_remap.restype = C.c_uint
        L.gjo_parse_channel_remap.argtypes = [C.c_char_p]
        L.gjo_channel_remap.argtypes = [C.POINTER(Image), u8p, C.c_uint]
        L.gjo_fill_noise.argtypes = [u8p, C.c_size_t, C.c_uint]
        L.gjo_fill_gradient.argtypes = [u8p, C.c_int, C.c_int, C.c_int]
        L.gjo_quant_table.argtypes = [C.c_int, C.c_int, u8p, C.POINTER(C.c_float), C.POINTER(C.c_uint16)]
        L.gjo_fdct_quant_block.argtypes = [u8p, C.c_int, C.POINTER(C.c_float), i16p]
        L.gjo_idct_block.argtypes = [i16p, C.POINTER(C.c_uint16), u8p, C.c_int]
        L.gjo_set_fma.argtypes = [C.c_int]
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i16(a):
    return a.ctypes.data_as(C.POINTER(C.c_int16))


def make_image(width, height, pixel_format=PF_444_P012, color_space=CS_RGB, quality=75, restart_interval=-1,
               interleaved=0, subsampling=None, color_space_internal=CS_BT601_256, width_padding=0, segment_info=0,
               header_type=0):
    """Build an Image the way gpujpeg_encoder_encode would see its parameters.
    subsampling: None (derive from pixel format) or list of (h, v) per component."""
    img = Image()
    img.width, img.height, img.width_padding = width, height, width_padding
    img.pixel_format, img.color_space = pixel_format, color_space
    img.quality, img.restart_interval, img.interleaved = quality, restart_interval, interleaved
    img.color_space_internal, img.segment_info, img.header_type = color_space_internal, segment_info, header_type
    if subsampling is not None:
        img.comp_count = len(subsampling)
        for i, (h, v) in enumerate(subsampling):
            img.samp_h[i], img.samp_v[i] = h, v
    lib().gjo_adjust_encoder_params(C.byref(img))
    if lib().gjo_image_init(C.byref(img)) != 0:
        raise ValueError("invalid image parameters")
    return img


def raw_size(width, height, pixel_format, width_padding=0):
    return lib().gjo_raw_size(width, height, width_padding, pixel_format)


def noise(n, seed=12345):
    a = np.empty(n, np.uint8)
    lib().gjo_fill_noise(_u8(a), n, seed)
    return a


def gradient(width, height, bpp):
    a = np.empty(width * height * bpp, np.uint8)
    lib().gjo_fill_gradient(_u8(a), width, height, bpp)
    return a


def preprocess(img, raw):
    raw = np.ascontiguousarray(raw, np.uint8)
    planes = np.empty(img.data_size, np.uint8)
    lib().gjo_preprocess(C.byref(img), _u8(raw), _u8(planes))
    return planes


def fdct_quant(img, planes):
    coefs = np.empty(img.data_size, np.int16)
    lib().gjo_fdct_quant(C.byref(img), _u8(planes), _i16(coefs))
    return coefs


def huffman_encode_segment(img, coefs, index):
    out = np.empty(img.block_count * 512 + 64 if img.restart_interval == 0 else 64 * 512 * max(1, img.restart_interval) + 64, np.uint8)
    n = lib().gjo_huffman_encode_segment(C.byref(img), _i16(coefs), index, _u8(out))
    return out[:n].copy()


def encode_from_coefs(img, coefs):
    cap = 4096 + int(img.data_size) * 4 + img.segment_count * 8
    out = np.empty(cap, np.uint8)
    n = lib().gjo_encode_from_coefs(C.byref(img), _i16(coefs), _u8(out), cap)
    if n == 0:
        raise RuntimeError("oracle encode overflow")
    return out[:n].copy()


def encode(img, raw):
    """raw -> complete JPEG bytes (numpy uint8)."""
    return encode_from_coefs(img, fdct_quant(img, preprocess(img, raw)))


def parse(jpeg, req_pixel_format=-1, req_color_space=-1):
    jpeg = np.ascontiguousarray(jpeg, np.uint8)
    s = Stream()
    if lib().gjo_parse(_u8(jpeg), jpeg.size, req_pixel_format, req_color_space, C.byref(s)) != 0:
        lib().gjo_stream_free(C.byref(s))
        raise ValueError("oracle: cannot parse JPEG")
    return s


def huffman_decode(stream, jpeg):
    jpeg = np.ascontiguousarray(jpeg, np.uint8)
    coefs = np.empty(stream.img.data_size, np.int16)
    lib().gjo_huffman_decode(C.byref(stream), _u8(jpeg), _i16(coefs))
    return coefs


def idct(stream, coefs):
    planes = np.empty(stream.img.data_size + 64, np.uint8)
    lib().gjo_idct(C.byref(stream), _i16(coefs), _u8(planes))
    return planes[:stream.img.data_size]


def postprocess(img, planes):
    raw = np.zeros(img.raw_size, np.uint8)
    planes = np.ascontiguousarray(planes, np.uint8)
    lib().gjo_postprocess(C.byref(img), _u8(planes), _u8(raw))
    return raw


def decode(jpeg, req_pixel_format=-1, req_color_space=-1):
    """JPEG bytes -> (raw pixels, Image). Defaults follow the reference: RGB 444-u8-p012 (u8 for grayscale)."""
    s = parse(jpeg, req_pixel_format, req_color_space)
    try:
        coefs = huffman_decode(s, jpeg)
        planes = idct(s, coefs)
        raw = postprocess(s.img, planes)
        img = Image.from_buffer_copy(s.img)
    finally:
        lib().gjo_stream_free(C.byref(s))
    return raw, img


def flip_planes(img, planes):
    """Vertical flip of the padded component planes (encoder option enc_opt_flipped / dec_opt_flipped)."""
    out = np.ascontiguousarray(planes).copy()
    lib().gjo_flip_planes(C.byref(img), _u8(out))
    return out


def channel_remap(img, raw, mapping):
    """Channel permutation of the raw image, e.g. "210" (enc_opt_channel_remap / dec_opt_channel_remap)."""
    m = lib().gjo_parse_channel_remap(mapping.encode())
    out = np.ascontiguousarray(raw).copy()
    if m == 0 or lib().gjo_channel_remap(C.byref(img), _u8(out), m) != 0:
        raise ValueError("invalid channel mapping for this pixel format")
    return out


def color_transform(cs_from, cs_to, rgb):
    c = (C.c_uint8 * 3)(*rgb)
    lib().gjo_color_transform(cs_from, cs_to, c)
    return tuple(c)


def have_ref():
    return os.path.exists(REF_PATH)
