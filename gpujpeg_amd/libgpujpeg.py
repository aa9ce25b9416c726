"""ctypes binding of the libgpujpeg C ABI (reference: libgpujpeg/gpujpeg_common.h, gpujpeg_encoder.h,
gpujpeg_decoder.h). Struct layouts mirror the reference headers field by field because they ARE the ABI.

The binding is library-agnostic: `Library(path)` loads any shared object exporting the libgpujpeg symbols.
The product library is gpujpeg_amd/lib/libgpujpeg.so (HIP kernels for gfx950); there is no CPU fallback --
loading fails loudly if it has not been built.
"""
import ctypes as C
import os

import numpy as np

MAX_COMPONENT_COUNT = 4

# enum gpujpeg_color_space (gpujpeg_type.h:85-94)
NONE, RGB, YCBCR_BT601, YCBCR_BT601_256LVLS, YCBCR_BT709, YUV = 0, 1, 2, 3, 4, 5
YCBCR_JPEG = YCBCR_BT601_256LVLS
CS_DEFAULT = -1
# enum gpujpeg_pixel_format (gpujpeg_type.h:108-134)
PIXFMT_NONE = -1
U8, P012_444, P0P1P2_444, P1020_422, P0P1P2_422, P0P1P2_420, P0123_4444 = 0, 1, 2, 3, 4, 5, 6
PIXFMT_AUTODETECT, PIXFMT_NO_ALPHA, PIXFMT_STD, PIXFMT_NATIVE = -2, -3, -4, -5
# enum gpujpeg_encoder_input_type / gpujpeg_decoder_output_type
ENCODER_INPUT_IMAGE, ENCODER_INPUT_OPENGL_TEXTURE, ENCODER_INPUT_GPU_IMAGE = 0, 1, 2
(DECODER_OUTPUT_INTERNAL_BUFFER, DECODER_OUTPUT_CUSTOM_BUFFER, DECODER_OUTPUT_OPENGL_TEXTURE,
 DECODER_OUTPUT_CUDA_BUFFER, DECODER_OUTPUT_CUSTOM_CUDA_BUFFER) = range(5)
RESTART_AUTO, RESTART_NONE = -1, 0


def MK_SUBSAMPLING(*f):
    f = list(f) + [0] * (8 - len(f))
    v = 0
    for x in f:
        v = (v << 4) | x
    return v


SUBSAMPLING_444 = MK_SUBSAMPLING(1, 1, 1, 1, 1, 1)
SUBSAMPLING_422 = MK_SUBSAMPLING(2, 1, 1, 1, 1, 1)
SUBSAMPLING_420 = MK_SUBSAMPLING(2, 2, 1, 1, 1, 1)
SUBSAMPLING_4444 = MK_SUBSAMPLING(1, 1, 1, 1, 1, 1, 1, 1)
SUBSAMPLING_400 = MK_SUBSAMPLING(1, 1)


class SamplingFactor(C.Structure):
    _fields_ = [("horizontal", C.c_uint8), ("vertical", C.c_uint8)]


class Parameters(C.Structure):  # struct gpujpeg_parameters, gpujpeg_common.h:176-215
    _fields_ = [("verbose", C.c_int), ("perf_stats", C.c_int), ("quality", C.c_int), ("restart_interval", C.c_int),
                ("interleaved", C.c_int), ("segment_info", C.c_int), ("comp_count", C.c_int),
                ("sampling_factor", SamplingFactor * MAX_COMPONENT_COUNT), ("color_space_internal", C.c_int)]


class ImageParameters(C.Structure):  # struct gpujpeg_image_parameters, gpujpeg_common.h:283-294
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("color_space", C.c_int), ("pixel_format", C.c_int),
                ("width_padding", C.c_int)]


class EncoderInput(C.Structure):  # gpujpeg_encoder.h:57-67
    _fields_ = [("type", C.c_int), ("image", C.c_void_p), ("texture", C.c_void_p)]


class DecoderOutput(C.Structure):  # gpujpeg_decoder.h:66-84
    _fields_ = [("type", C.c_int), ("data", C.c_void_p), ("data_size", C.c_size_t), ("param_image", ImageParameters),
                ("texture", C.c_void_p), ("metadata", C.c_void_p)]


class DecoderInitParameters(C.Structure):  # gpujpeg_decoder.h:90-97
    _fields_ = [("stream", C.c_void_p), ("verbose", C.c_int), ("perf_stats", C.c_bool), ("ff_cs_itu601_is_709", C.c_bool)]


class DurationStats(C.Structure):  # gpujpeg_common.h:352-362
    _fields_ = [(n, C.c_double) for n in ("duration_memory_to", "duration_memory_from", "duration_memory_map",
                                          "duration_memory_unmap", "duration_preprocessor", "duration_dct_quantization",
                                          "duration_huffman_coder", "duration_stream", "duration_in_gpu")]


class _ImageInfoFields(C.Structure):
    _fields_ = [("param_image", ImageParameters), ("param", Parameters), ("segment_count", C.c_int),
                ("header_type", C.c_int), ("comment", C.c_char_p), ("metadata", C.c_uint8 * 8)]


class ImageInfo(C.Union):  # gpujpeg_decoder.h:270-283 (512-byte union)
    _anonymous_ = ("f",)
    _fields_ = [("f", _ImageInfoFields), ("reserved", C.c_char * 512)]


PRODUCT_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgpujpeg.so")


class Library:
    """Thin typed handle on a libgpujpeg shared object."""

    def __init__(self, path=None):
        path = path or PRODUCT_LIB
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(the HIP library is mandatory, there is no CPU fallback)")
        self.path = path
        L = self.L = C.CDLL(path, mode=C.RTLD_LOCAL)
        vp = C.c_void_p
        L.gpujpeg_set_default_parameters.argtypes = [C.POINTER(Parameters)]
        L.gpujpeg_image_set_default_parameters.argtypes = [C.POINTER(ImageParameters)]
        L.gpujpeg_parameters_chroma_subsampling.argtypes = [C.POINTER(Parameters), C.c_uint32]
        L.gpujpeg_image_calculate_size.restype = C.c_size_t
        L.gpujpeg_image_calculate_size.argtypes = [C.POINTER(ImageParameters)]
        L.gpujpeg_init_device.argtypes = [C.c_int, C.c_int]
        L.gpujpeg_encoder_create.restype = vp
        L.gpujpeg_encoder_create.argtypes = [vp]
        L.gpujpeg_encoder_destroy.argtypes = [vp]
        L.gpujpeg_encoder_encode.argtypes = [vp, C.POINTER(Parameters), C.POINTER(ImageParameters), C.POINTER(EncoderInput),
                                             C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.gpujpeg_encoder_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.gpujpeg_encoder_suggest_restart_interval.argtypes = [C.POINTER(ImageParameters), C.c_uint32, C.c_bool, C.c_int]
        L.gpujpeg_encoder_get_stats.argtypes = [vp, C.POINTER(DurationStats)]
        L.gpujpeg_decoder_create.restype = vp
        L.gpujpeg_decoder_create.argtypes = [vp]
        L.gpujpeg_decoder_destroy.argtypes = [vp]
        L.gpujpeg_decoder_init.argtypes = [vp, C.POINTER(Parameters), C.POINTER(ImageParameters)]
        L.gpujpeg_decoder_decode.argtypes = [vp, vp, C.c_size_t, C.POINTER(DecoderOutput)]
        L.gpujpeg_decoder_set_output_format.argtypes = [vp, C.c_int, C.c_int]
        L.gpujpeg_decoder_set_output_format.restype = None
        L.gpujpeg_decoder_get_stats.argtypes = [vp, C.POINTER(DurationStats)]
        L.gpujpeg_decoder_get_image_info.argtypes = [vp, C.c_size_t, C.POINTER(ImageParameters), C.POINTER(Parameters), C.POINTER(C.c_int)]
        L.gpujpeg_decoder_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.gpujpeg_version.restype = C.c_int
        if hasattr(L, "gpujpeg_amd_encoder_read_coefficients"):  # MI355X extensions (include/gpujpeg_amd_ext.h)
            for n in ("gpujpeg_amd_encoder_read_coefficients", "gpujpeg_amd_decoder_read_coefficients"):
                getattr(L, n).restype = C.c_size_t
                getattr(L, n).argtypes = [vp, C.POINTER(C.c_int16), C.c_size_t]
            for n in ("gpujpeg_amd_encoder_read_planes", "gpujpeg_amd_decoder_read_planes"):
                getattr(L, n).restype = C.c_size_t
                getattr(L, n).argtypes = [vp, C.POINTER(C.c_uint8), C.c_size_t]
            for n in ("gpujpeg_amd_encoder_get_kernel_times", "gpujpeg_amd_decoder_get_kernel_times"):
                getattr(L, n).argtypes = [vp, C.POINTER(C.c_float)]
            for n in ("gpujpeg_amd_encoder_set_fused", "gpujpeg_amd_decoder_set_fused", "gpujpeg_amd_decoder_keep_coefficients", "gpujpeg_amd_encoder_keep_coefficients"):
                getattr(L, n).restype = None
                getattr(L, n).argtypes = [vp, C.c_int]
        if hasattr(L, "gpujpeg_amd_encoder_encode_batch"):  # frame batches (include/gpujpeg_amd_ext.h)
            L.gpujpeg_amd_encoder_encode_batch.argtypes = [vp, C.POINTER(Parameters), C.POINTER(ImageParameters), vp, C.c_size_t, C.c_int,
                                                           C.POINTER(vp), C.POINTER(C.c_size_t)]
        for n in ("gpujpeg_amd_encoder_set_batch_chunk", "gpujpeg_amd_decoder_set_batch_chunk"):
            if hasattr(L, n):
                getattr(L, n).restype = None
                getattr(L, n).argtypes = [vp, C.c_int]
        for n in ("gpujpeg_amd_encoder_last_batch", "gpujpeg_amd_decoder_last_batch"):
            if hasattr(L, n):
                getattr(L, n).argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if hasattr(L, "gpujpeg_amd_encoder_encode_batch_ptrs"):
            L.gpujpeg_amd_encoder_encode_batch_ptrs.argtypes = [vp, C.POINTER(Parameters), C.POINTER(ImageParameters), C.POINTER(vp), C.c_int,
                                                                C.POINTER(vp), C.POINTER(C.c_size_t)]
            L.gpujpeg_amd_decoder_decode_batch_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_int, C.POINTER(vp), C.c_size_t,
                                                                C.POINTER(ImageParameters)]
        if hasattr(L, "gpujpeg_amd_decoder_decode_batch"):
            L.gpujpeg_amd_decoder_decode_batch.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, vp, C.c_size_t, C.POINTER(ImageParameters)]

    # ---- developer settings (include/gpujpeg_amd_ext.h: gpujpeg_amd_tuning) ----
    def tuning(self, setting):
        """"NAME=VALUE" / "NAME" for the coders created from now on, None forgets every setting. True when the library took it
        (False: unknown name, or a library without the call -- the reference builds)."""
        if not hasattr(self.L, "gpujpeg_amd_tuning"):
            return False
        self.L.gpujpeg_amd_tuning.argtypes = [C.c_char_p]
        return self.L.gpujpeg_amd_tuning(None if setting is None else setting.encode()) == 0

    def tuning_names(self):
        if not hasattr(self.L, "gpujpeg_amd_tuning_names"):
            return []
        self.L.gpujpeg_amd_tuning_names.restype = C.POINTER(C.c_char_p)
        arr, out, i = self.L.gpujpeg_amd_tuning_names(), [], 0
        while arr[i]:
            out.append(arr[i].decode())
            i += 1
        return out

    # ---- parameter helpers ----
    def default_parameters(self):
        p = Parameters()
        self.L.gpujpeg_set_default_parameters(C.byref(p))
        return p

    def default_image_parameters(self):
        p = ImageParameters()
        self.L.gpujpeg_image_set_default_parameters(C.byref(p))
        return p

    def image_size(self, param_image):
        return self.L.gpujpeg_image_calculate_size(C.byref(param_image))


def apply_environment_settings(lib, environ=None):
    """DEVELOPER AID for tests and measurement tools: hand the developer settings named in the environment (GJ_DEC_TOKENS=1, GJ_ENC_TAIL=0, ...;
    INTEGRATION.md) to the library through gpujpeg_amd_tuning -- the library itself never reads the environment. Coders created afterwards
    take them. No-op for libraries without the call (the reference builds under oracle/_ref)."""
    import os
    env = os.environ if environ is None else environ
    if not lib.tuning(None):
        return
    for name in lib.tuning_names():
        if name in env:
            lib.tuning(f"{name}={env[name]}")


class Encoder:
    """Mirror of the reference encoder object: gpujpeg_encoder_create/encode/destroy (gpujpeg_encoder.h:118-176)."""

    def __init__(self, lib, stream=None):
        self.lib = lib
        self.h = lib.L.gpujpeg_encoder_create(stream)
        if not self.h:
            raise RuntimeError("gpujpeg_encoder_create failed")

    def set_option(self, opt, val):
        return self.lib.L.gpujpeg_encoder_set_option(self.h, opt.encode(), val.encode())

    def encode(self, param, param_image, image, gpu=False):
        """image: numpy uint8 array (host) or an integer device pointer when gpu=True. Returns numpy uint8 copy."""
        ptr, n = self.encode_noclone(param, param_image, image, gpu)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    def encode_noclone(self, param, param_image, image, gpu=False):
        inp = EncoderInput()
        if gpu:
            inp.type, inp.image = ENCODER_INPUT_GPU_IMAGE, int(image)
        else:
            image = np.ascontiguousarray(image, np.uint8)
            self._keep = image
            inp.type, inp.image = ENCODER_INPUT_IMAGE, image.ctypes.data
        out, size = C.POINTER(C.c_uint8)(), C.c_size_t(0)
        rc = self.lib.L.gpujpeg_encoder_encode(self.h, C.byref(param), C.byref(param_image), C.byref(inp), C.byref(out), C.byref(size))
        if rc != 0:
            raise RuntimeError(f"gpujpeg_encoder_encode failed ({rc})")
        return out, size.value

    def encode_batch_noclone(self, param, param_image, frames, count, stride=None, gpu=False):
        """gpujpeg_amd_encoder_encode_batch: `count` frames of one geometry behind one set of launches. frames: numpy uint8 array holding
        the frames back to back (host), or an integer device pointer when gpu=True; stride: bytes between two frames (default: a frame).
        Returns ([pointer], [size]) of the streams -- device pointers with enc_opt_out=device, host pointers otherwise."""
        if gpu:
            base = int(frames)
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            self._keep = frames
            base = frames.ctypes.data
        if stride is None:
            stride = self.lib.image_size(param_image)
        ptrs, sizes = (C.c_void_p * count)(), (C.c_size_t * count)()
        rc = self.lib.L.gpujpeg_amd_encoder_encode_batch(self.h, C.byref(param), C.byref(param_image), base, stride, count, ptrs, sizes)
        if rc != 0:
            raise RuntimeError(f"gpujpeg_amd_encoder_encode_batch failed ({rc})")
        return [int(p or 0) for p in ptrs], [int(n) for n in sizes]

    def set_batch_chunk(self, frames):
        self.lib.L.gpujpeg_amd_encoder_set_batch_chunk(self.h, int(frames))

    def last_batch(self):
        """(frames coded by the batched launches, frames coded one by one) of the last encode_batch call"""
        a, b = C.c_int(0), C.c_int(0)
        self.lib.L.gpujpeg_amd_encoder_last_batch(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def encode_batch_ptrs(self, param, param_image, frames):
        """gpujpeg_amd_encoder_encode_batch_ptrs: frames = list of numpy uint8 arrays (host) and / or integer device pointers, one buffer per frame.
        Returns the streams as numpy copies (encoder output in host memory, the default)."""
        n = len(frames)
        keep = [np.ascontiguousarray(f, np.uint8) if not isinstance(f, int) else f for f in frames]
        self._keep = keep
        fp = (C.c_void_p * n)(*[f if isinstance(f, int) else f.ctypes.data for f in keep])
        ptrs, sizes = (C.c_void_p * n)(), (C.c_size_t * n)()
        rc = self.lib.L.gpujpeg_amd_encoder_encode_batch_ptrs(self.h, C.byref(param), C.byref(param_image), fp, n, ptrs, sizes)
        if rc != 0:
            raise RuntimeError(f"gpujpeg_amd_encoder_encode_batch_ptrs failed ({rc})")
        return [np.frombuffer((C.c_uint8 * int(s)).from_address(int(p)), np.uint8).copy() for p, s in zip(ptrs, sizes)]

    def encode_batch(self, param, param_image, frames, count, stride=None):
        """host frames in, list of numpy uint8 copies of the streams out (encoder output in host memory, the default)"""
        ptrs, sizes = self.encode_batch_noclone(param, param_image, frames, count, stride)
        return [np.frombuffer((C.c_uint8 * n).from_address(p), np.uint8).copy() for p, n in zip(ptrs, sizes)]

    def stats(self):
        s = DurationStats()
        self.lib.L.gpujpeg_encoder_get_stats(self.h, C.byref(s))
        return s

    def set_fused(self, enabled):
        self.lib.L.gpujpeg_amd_encoder_set_fused(self.h, int(enabled))

    def keep_coefficients(self, enabled=True):
        """Leave the quantised coefficients of the following encode calls in HBM (for coefficients())."""
        self.lib.L.gpujpeg_amd_encoder_keep_coefficients(self.h, int(enabled))

    def kernel_times(self):
        ms = (C.c_float * 8)()
        if self.lib.L.gpujpeg_amd_encoder_get_kernel_times(self.h, ms) != 0:
            return None
        return list(ms)[:5]

    def coefficients(self, count):
        a = np.empty(count, np.int16)
        n = self.lib.L.gpujpeg_amd_encoder_read_coefficients(self.h, a.ctypes.data_as(C.POINTER(C.c_int16)), count)
        return a[:n]

    def planes(self, count):
        a = np.empty(count, np.uint8)
        n = self.lib.L.gpujpeg_amd_encoder_read_planes(self.h, a.ctypes.data_as(C.POINTER(C.c_uint8)), count)
        return a[:n]

    def close(self):
        if self.h:
            self.lib.L.gpujpeg_encoder_destroy(self.h)
            self.h = None

    __del__ = close


class Decoder:
    """Mirror of the reference decoder object: gpujpeg_decoder_create/decode/destroy (gpujpeg_decoder.h:153-202)."""

    def __init__(self, lib, stream=None):
        self.lib = lib
        self.h = lib.L.gpujpeg_decoder_create(stream)
        if not self.h:
            raise RuntimeError("gpujpeg_decoder_create failed")

    def set_output_format(self, color_space, pixel_format):
        self.lib.L.gpujpeg_decoder_set_output_format(self.h, color_space, pixel_format)

    def init(self, param, param_image):
        return self.lib.L.gpujpeg_decoder_init(self.h, C.byref(param), C.byref(param_image))

    def decode(self, jpeg, device_out=None):
        """jpeg: numpy uint8 array. Returns (numpy uint8 copy of pixels, ImageParameters); with device_out
        (integer device pointer) decodes into that buffer and returns (None, ImageParameters)."""
        jpeg = np.ascontiguousarray(jpeg, np.uint8)
        out = DecoderOutput()
        if device_out is not None:
            out.type, out.data = DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, int(device_out)
        else:
            out.type = DECODER_OUTPUT_INTERNAL_BUFFER
        rc = self.lib.L.gpujpeg_decoder_decode(self.h, jpeg.ctypes.data, jpeg.size, C.byref(out))
        if rc != 0:
            raise RuntimeError(f"gpujpeg_decoder_decode failed ({rc})")
        if device_out is not None:
            return None, out.param_image
        buf = (C.c_uint8 * out.data_size).from_address(out.data)
        return np.frombuffer(buf, np.uint8).copy(), out.param_image

    def set_batch_chunk(self, frames):
        self.lib.L.gpujpeg_amd_decoder_set_batch_chunk(self.h, int(frames))

    def last_batch(self):
        """(frames decoded by the batched launches, frames decoded one by one) of the last decode_batch call"""
        a, b = C.c_int(0), C.c_int(0)
        self.lib.L.gpujpeg_amd_decoder_last_batch(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def decode_batch_ptrs(self, streams, frame_bytes):
        """gpujpeg_amd_decoder_decode_batch_ptrs: one buffer per stream and per decoded frame (host memory here). Returns (list of pixel arrays, ImageParameters)."""
        n = len(streams)
        keep = [np.ascontiguousarray(s, np.uint8) for s in streams]
        outs = [np.empty(frame_bytes, np.uint8) for _ in range(n)]
        sp = (C.c_void_p * n)(*[s.ctypes.data for s in keep])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        csz = (C.c_size_t * n)(*[s.size for s in keep])
        pi = ImageParameters()
        rc = self.lib.L.gpujpeg_amd_decoder_decode_batch_ptrs(self.h, sp, csz, n, op, frame_bytes, C.byref(pi))
        if rc != 0:
            raise RuntimeError(f"gpujpeg_amd_decoder_decode_batch_ptrs failed ({rc})")
        raw = self.lib.image_size(pi)
        return [o[:raw] for o in outs], pi

    def decode_batch(self, streams, device_out=None, out_stride=None, device_in=None, in_stride=None, sizes=None, frame_bytes=None):
        """gpujpeg_amd_decoder_decode_batch: streams with one header behind one set of launches. streams: list of numpy uint8 arrays (host;
        packed into one buffer here), or device_in = integer device pointer of stream 0 with in_stride and sizes. Pixels go to device_out
        (integer device pointer, frames out_stride apart) or come back as a list of numpy arrays. Returns (pixels or None, ImageParameters)."""
        if device_in is None:
            sizes = [int(x.size) for x in streams]
            in_stride = (max(sizes) + 64 + 15) & ~15
            buf = np.zeros(in_stride * len(sizes), np.uint8)
            for i, x in enumerate(streams):
                buf[i * in_stride:i * in_stride + x.size] = x
            self._keep = buf
            base = buf.ctypes.data
        else:
            base = int(device_in)
        n = len(sizes)
        csz = (C.c_size_t * n)(*sizes)
        pi = ImageParameters()
        if device_out is not None:
            rc = self.lib.L.gpujpeg_amd_decoder_decode_batch(self.h, base, in_stride, csz, n, int(device_out), out_stride, C.byref(pi))
            if rc != 0:
                raise RuntimeError(f"gpujpeg_amd_decoder_decode_batch failed ({rc})")
            return None, pi
        # host output: the frame size is not known before the first stream has been parsed -- ask the library (streams in host memory),
        # or take the caller's word (device streams: frame_bytes = room per decoded frame)
        if device_in is not None:
            if not frame_bytes:
                raise ValueError("decode_batch(device_in=..., device_out=None) needs frame_bytes: the size of a decoded frame is not known "
                                 "before a stream has been parsed, and the streams are in device memory")
            bound = int(frame_bytes)
        else:
            pi0, p0 = ImageParameters(), Parameters()
            first = np.ascontiguousarray(streams[0])
            if self.lib.L.gpujpeg_decoder_get_image_info(first.ctypes.data_as(C.c_void_p), first.size, C.byref(pi0), C.byref(p0), None) != 0:
                raise RuntimeError("gpujpeg_decoder_get_image_info failed")
            bound = max(int(pi0.width) * int(pi0.height) * 4 + 4096, 1)  # (the library checks the real frame size against the stride)
        out = np.empty(bound * n, np.uint8)
        rc = self.lib.L.gpujpeg_amd_decoder_decode_batch(self.h, base, in_stride, csz, n, out.ctypes.data, bound, C.byref(pi))
        if rc != 0:
            raise RuntimeError(f"gpujpeg_amd_decoder_decode_batch failed ({rc})")
        raw = self.lib.image_size(pi)
        return [out[i * bound:i * bound + raw].copy() for i in range(n)], pi

    def path_counters(self):
        """(speculative launches, of those without the k_marker_table launch, decoded again the careful way) of this decoder so far"""
        a = (C.c_long * 3)()
        self.lib.L.gpujpeg_amd_decoder_get_path_counters.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        assert self.lib.L.gpujpeg_amd_decoder_get_path_counters(self.h, a) == 0
        return tuple(int(x) for x in a)

    def stats(self):
        s = DurationStats()
        self.lib.L.gpujpeg_decoder_get_stats(self.h, C.byref(s))
        return s

    def set_fused(self, enabled):
        self.lib.L.gpujpeg_amd_decoder_set_fused(self.h, int(enabled))

    def keep_coefficients(self, enabled=True):
        """Leave the quantised coefficients of the following decode calls in HBM (for coefficients())."""
        self.lib.L.gpujpeg_amd_decoder_keep_coefficients(self.h, int(enabled))

    def kernel_times(self):
        ms = (C.c_float * 8)()
        if self.lib.L.gpujpeg_amd_decoder_get_kernel_times(self.h, ms) != 0:
            return None
        return list(ms)[:4]

    def coefficients(self, count):
        a = np.empty(count, np.int16)
        n = self.lib.L.gpujpeg_amd_decoder_read_coefficients(self.h, a.ctypes.data_as(C.POINTER(C.c_int16)), count)
        return a[:n]

    def planes(self, count):
        a = np.empty(count, np.uint8)
        n = self.lib.L.gpujpeg_amd_decoder_read_planes(self.h, a.ctypes.data_as(C.POINTER(C.c_uint8)), count)
        return a[:n]

    def close(self):
        if self.h:
            self.lib.L.gpujpeg_decoder_destroy(self.h)
            self.h = None

    __del__ = close
