import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (oracle/gj_oracle.c through ctypes); built on demand."""
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def G():
    """The ctypes binding. The tests force kernel paths with the developer settings of INTEGRATION.md by name (monkeypatch.setenv("GJ_DEC_TOKENS", "1")),
    and the release library does not read the environment (round 6): every coder the tests create hands the settings named in the environment at that
    moment to the library first (gpujpeg_amd_tuning; libgpujpeg.apply_environment_settings)."""
    from gpujpeg_amd import libgpujpeg
    if not getattr(libgpujpeg, "_settings_from_environment", False):
        for cls in (libgpujpeg.Encoder, libgpujpeg.Decoder):
            def patched(self, lib, *a, _orig=cls.__init__, **kw):
                if libgpujpeg._settings_from_environment:  # (a test of the settings call itself switches this off for its duration)
                    libgpujpeg.apply_environment_settings(lib)
                _orig(self, lib, *a, **kw)
            cls.__init__ = patched
        libgpujpeg._settings_from_environment = True
    return libgpujpeg


@pytest.fixture(scope="session")
def _ref_lib(O, G):
    if not O.have_ref():
        pytest.skip("oracle/_ref/libgpujpeg_ref.so not built (needs /root/reference)")
    return G.Library(O.REF_PATH)


@pytest.fixture
def ref(O, _ref_lib):
    """The reference's own code on the CPU (oracle/_ref/libgpujpeg_ref.so): its host C files and its CUDA translation units for
    colour/sampling, fDCT+quantisation and dequantisation+IDCT, compiled where they lie with contraction OFF (oracle/Makefile).
    While a test holds this fixture the restatement runs in the same mode (gjo_set_fma(0)), so the two must agree bit for bit;
    the fused map is pinned on the GPU box (tests/test_gpu_refhip.py). Absent on the GPU box only if it was not prebuilt."""
    O.lib().gjo_set_fma(0)
    yield _ref_lib
    O.lib().gjo_set_fma(1)


@pytest.fixture(scope="session")
def lib(G):
    """The product: gpujpeg_amd/lib/libgpujpeg.so. Missing library = hard failure, never a fallback."""
    import torch  # noqa: F401  -- before the library: PyTorch bundles its own libamdhip64; whichever HIP runtime is mapped first
    #                              serves the whole process, and a second copy initialised later finds no device
    if not os.path.exists(G.PRODUCT_LIB):
        import __graft_entry__
        __graft_entry__.build()
    return G.Library()


@pytest.fixture(scope="session")
def gpu_lib(lib):
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    assert lib.L.gpujpeg_init_device(0, 0) == 0
    return lib


def natural_image(w, h, c=3, seed=1):
    """Smooth structure + noise, compresses like a photograph."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    chans = [128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0), xx * 255.0 / max(w, 1), yy * 255.0 / max(h, 1), 128 + 0.0 * xx][:c]
    img = np.stack(chans, -1) + rng.normal(0, 6, (h, w, c))
    return np.clip(img, 0, 255).astype(np.uint8).reshape(-1)


def psnr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = np.mean((a - b) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


# (name, width, height, pixel format, colour space, quality, restart, interleaved, subsampling [(h,v)...] or None, internal cs)
# pixel formats / colour spaces use the reference's enum values
CASES = [
    ("rgb_tiny_r4", 64, 64, 1, 1, 75, 4, 0, None, 3),
    ("rgb_natural_auto", 640, 368, 1, 1, 75, -1, 0, None, 3),
    ("rgb_odd_noise", 119, 61, 1, 1, 75, -1, 0, None, 3),
    ("rgb_hdlike_r24", 480, 272, 1, 1, 75, 24, 0, None, 3),
    ("rgb_interleaved", 333, 123, 1, 1, 75, -1, 1, None, 3),
    ("rgb_q100", 96, 80, 1, 1, 100, 8, 0, None, 3),
    ("rgb_q1", 96, 80, 1, 1, 1, 8, 0, None, 3),
    ("rgb_q50_r1", 40, 24, 1, 1, 50, 1, 0, None, 3),
    ("rgb_restart0", 100, 60, 1, 1, 75, 0, 0, None, 3),
    ("rgb_big_restart", 512, 256, 1, 1, 75, 300, 0, None, 3),
    ("rgb_1x1", 1, 1, 1, 1, 75, 8, 0, None, 3),
    ("rgb_7x9", 7, 9, 1, 1, 75, 2, 1, None, 3),
    ("uyvy_422_il_q90", 322, 50, 3, 3, 90, -1, 1, None, 3),
    # NB: odd widths with the packed 4:2:2 format are not a valid case: the reference rounds the width up to even inside the
    # kernel only (src/gpujpeg_preprocessor.cu:369-373) while the buffer size stays odd, i.e. it reads past the caller's buffer.
    ("rgb_to_420_il", 322, 242, 1, 1, 75, -1, 1, [(2, 2), (1, 1), (1, 1)], 3),
    ("rgb_to_422_nonil", 200, 100, 1, 1, 80, 5, 0, [(2, 1), (1, 1), (1, 1)], 3),
    ("planar420_in", 162, 122, 5, 3, 75, -1, 0, None, 3),
    ("planar422_in", 163, 90, 4, 3, 75, -1, 1, None, 3),
    ("planar444_in", 160, 90, 2, 3, 75, 7, 0, None, 3),
    ("gray", 333, 111, 0, 3, 75, -1, 0, None, 3),
    ("rgba_4444", 200, 100, 6, 1, 75, -1, 1, [(1, 1)] * 4, 3),
    ("rgb_internal_rgb", 120, 90, 1, 1, 75, 6, 0, None, 1),
    ("rgb_bt709", 120, 90, 1, 1, 75, 6, 1, None, 4),
    ("rgb_bt601", 120, 90, 1, 1, 75, 6, 0, None, 2),
    ("ycbcr709_to_jpeg", 120, 90, 1, 4, 75, 6, 0, None, 3),
]


def make_raw(O, case):
    name, w, h, pf, cs = case[:5]
    n = O.raw_size(w, h, pf)
    if "natural" in name or "hdlike" in name or "420" in name and pf == 1:
        comps = {0: 1, 1: 3, 6: 4}.get(pf)
        if comps:
            return natural_image(w, h, comps, seed=len(name))
    if "restart0" in name or "big_restart" in name:
        return natural_image(w, h, 3, seed=7)
    return O.noise(n, seed=12345 + len(name))


def oracle_image(O, case, **kw):
    name, w, h, pf, cs, q, ri, il, ss, csi = case
    return O.make_image(w, h, pixel_format=pf, color_space=cs, quality=q, restart_interval=ri, interleaved=il, subsampling=ss,
                        color_space_internal=csi, **kw)


def api_params(lib, G, case, segment_info=0):
    import ctypes as C
    name, w, h, pf, cs, q, ri, il, ss, csi = case
    p = lib.default_parameters()
    p.quality, p.restart_interval, p.interleaved, p.verbose, p.segment_info, p.color_space_internal = q, ri, il, -1, segment_info, csi
    if ss is not None:
        f = [x for hv in ss for x in hv]
        lib.L.gpujpeg_parameters_chroma_subsampling(C.byref(p), G.MK_SUBSAMPLING(*f))
    pi = lib.default_image_parameters()
    pi.width, pi.height, pi.pixel_format, pi.color_space = w, h, pf, cs
    return p, pi


def random_case(seed):
    """A random but valid configuration: pixel format, colour spaces, chroma sampling, size, quality, restart interval, interleaving
    (shared by the oracle-vs-reference and product-vs-oracle differential tests)."""
    rng = np.random.default_rng(9000 + seed)
    pf = int(rng.choice([0, 1, 1, 1, 2, 3, 4, 5, 6]))
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 160))
    q = int(rng.choice([1, 25, 50, 75, 90, 100, int(rng.integers(1, 101))]))
    ri = int(rng.choice([-1, -1, 0, 1, 2, 3, 5, 8, 13, 36, 100, int(rng.integers(1, 41))]))
    il = int(rng.integers(0, 2))
    ss, cs, csi = None, 3, 3
    if pf == 0:
        il = 0
    elif pf == 1:
        cs = int(rng.choice([1, 1, 1, 3, 4]))
        csi = int(rng.choice([3, 3, 2, 4, 1])) if cs == 1 else 3
        ss = [None, None, [(2, 2), (1, 1), (1, 1)], [(2, 1), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)]][int(rng.integers(0, 6))]
        if csi == 1:
            ss = None
    elif pf == 3:
        w += w & 1  # packed 4:2:2 needs an even width (see CASES)
        w = max(w, 2)
    elif pf == 6:
        cs, il, ss = 1, 1, [(1, 1)] * 4
    return (f"rand{seed}_pf{pf}", w, h, pf, cs, q, ri, il, ss, csi)


def random_raw(O, case, seed):
    name, w, h, pf, cs, q = case[:6]
    n = O.raw_size(w, h, pf)
    if q >= 95:
        # ramps with a little noise: incompressible data at q100 overflows the reference's own output buffer (1000 + 2 bytes per raw
        # sample, src/gpujpeg_encoder.c) once the padding to whole MCUs adds enough blocks -- its bug, not a parity case
        return ((np.arange(n, dtype=np.int64) * 3 // 11 + (O.noise(n, seed=seed) & 3)) % 256).astype(np.uint8)
    comps = {0: 1, 1: 3, 6: 4}.get(pf)
    return natural_image(w, h, comps, seed=seed) if comps and seed % 2 == 0 else O.noise(n, seed=777 + seed)
