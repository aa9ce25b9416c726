"""-m "not gpu": the product's OWN kernels (gpujpeg_amd/csrc/*.hip, unmodified) executed on the CPU by tests/hipemu -- a fiber per
work-item, wave64 cross-lane operations with the ISA's semantics, LDS, barriers -- behind the same public C API, against the same oracle
and with the same test bodies as the GPU tier (tests/test_gpu_parity.py; imported, not copied). What this tier can and cannot say:
it checks the LOGIC of every kernel (indexing, prefix sums, bit packing, DPP / ballot protocols, LDS bounds when built with
-fsanitize=address) bit for bit on small frames in a container without a GPU; it says nothing about timing, and the float stages run
on x86 FMA units instead of gfx950's (same IEEE operations, explicit fmaf only). The GPU tier remains the parity proof.
hipemu is test infrastructure: nothing under tests/ is linked into or loaded by the product."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import test_gpu_api as A
import test_gpu_parity as T
from conftest import CASES

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "hipemu")
EMU_LIB = os.path.join(EMU_DIR, "_build", "libgpujpeg_emu.so")


@pytest.fixture(scope="session")
def emu_lib(G):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or shutil.which("make") is None:
        pytest.skip("hipemu needs ROCm's clang++ (host compilation of the .hip files)")
    import fcntl
    with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:  # (pytest-xdist: one worker builds, the others wait for it)
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-s", "-j8", "-C", EMU_DIR], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    lib = G.Library(os.environ.get("GJ_EMU_LIB") or EMU_LIB)  # (GJ_EMU_LIB: another build of the execution model, e.g. the ASan one with its runtime preloaded)
    assert lib.L.gpujpeg_init_device(0, 0) == 0
    return lib


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "generic"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_emu_encode_decode_bit_exact(O, G, emu_lib, case, fused):
    T.test_encode_decode_bit_exact(O, G, emu_lib, case, fused)


@pytest.mark.parametrize("case", [c for c in CASES if c[6] != 0][:3], ids=lambda c: c[0])
def test_emu_segment_info(O, G, emu_lib, case):
    T.test_segment_info(O, G, emu_lib, case)


@pytest.mark.parametrize("pf,w,h", [(1, 161, 121), (3, 322, 77), (5, 33, 35), (0, 99, 3), (6, 17, 9)])
def test_emu_output_formats(O, G, emu_lib, pf, w, h):
    T.test_output_formats(O, G, emu_lib, pf, w, h)


@pytest.mark.parametrize("mode", ["par", "serial", "seq"])
@pytest.mark.parametrize("ec", T.ENTROPY_CASES, ids=[c[0] for c in T.ENTROPY_CASES])
def test_emu_entropy_decoder_variants(O, G, emu_lib, ec, mode, monkeypatch):
    T.test_entropy_decoder_variants(O, G, emu_lib, ec, mode, monkeypatch)


@pytest.mark.parametrize("tc", T.TOKEN_CASES, ids=[c[0] for c in T.TOKEN_CASES])
def test_emu_token_mode_decoder(O, G, emu_lib, tc, monkeypatch):
    T.test_token_mode_decoder(O, G, emu_lib, tc, monkeypatch)


@pytest.mark.parametrize("kernel", ["tok", "par"])
@pytest.mark.parametrize("fc", T.FOLD_CASES, ids=[c[0] for c in T.FOLD_CASES])
def test_emu_token_decoder_without_the_table_launch(O, G, emu_lib, fc, monkeypatch, kernel):
    T.test_token_decoder_without_the_table_launch(O, G, emu_lib, fc, monkeypatch, kernel)


def test_emu_token_mode_longer_sub_sequences(O, G, emu_lib, monkeypatch):
    T.test_token_mode_longer_sub_sequences(O, G, emu_lib, monkeypatch)


def test_emu_token_mode_tiny_segments(O, G, emu_lib, monkeypatch):
    T.test_token_mode_tiny_segments(O, G, emu_lib, monkeypatch)


def test_emu_token_mode_damaged_streams(O, G, emu_lib, monkeypatch):
    T.test_token_mode_damaged_streams(O, G, emu_lib, monkeypatch)


@pytest.mark.parametrize("tc", T.TOKEN_422_CASES, ids=[c[0] for c in T.TOKEN_422_CASES])
def test_emu_token_mode_decoder_422(O, G, emu_lib, tc, monkeypatch):
    T.test_token_mode_decoder_422(O, G, emu_lib, tc, monkeypatch)


@pytest.mark.parametrize("w,h,restart", [(648, 50, 6), (322, 77, 1), (640, 64, 64), (640, 64, 65), (16, 8, 3)])
def test_emu_packed_422_whole_frame_encoder(O, G, emu_lib, w, h, restart):
    T.test_packed_422_whole_frame_encoder(O, G, emu_lib, w, h, restart)


@pytest.mark.parametrize("tc", T.TAIL_CASES, ids=[c[0] for c in T.TAIL_CASES])
def test_emu_encoder_tiles_and_gather(O, G, emu_lib, tc, monkeypatch):
    T.test_encoder_tiles_and_gather(O, G, emu_lib, tc, monkeypatch)


@pytest.mark.parametrize("shape", [101, 103, 204, 401, 802, 1601, 1604])
def test_emu_marker_scan_shapes(O, G, emu_lib, shape, monkeypatch):
    T.test_marker_scan_shapes(O, G, emu_lib, shape, monkeypatch)


@pytest.mark.parametrize("w,h,ri", [(640, 64, 5), (320, 40, 1)])
def test_emu_dense_two_bit_tokens_all_decoder_paths(O, G, emu_lib, w, h, ri, monkeypatch):
    T.test_dense_two_bit_tokens_all_decoder_paths(O, G, emu_lib, w, h, ri, monkeypatch)


@pytest.mark.parametrize("layout", ["rgb444", "uyvy422"])
def test_emu_damaged_streams_token_mode_equals_plane_mode(O, G, emu_lib, layout, monkeypatch):
    T.test_damaged_streams_token_mode_equals_plane_mode(O, G, emu_lib, layout, monkeypatch)


def test_emu_reuse_padding_and_reconfiguration(O, G, emu_lib):
    T.test_width_padding(O, G, emu_lib)
    T.test_decoder_reuse_without_clearing(O, G, emu_lib)
    T.test_zero_image_round_trip(O, G, emu_lib)
    T.test_encoder_path_changes_between_frames(O, G, emu_lib)


@pytest.mark.parametrize("seed", range(0, 160, 4))
def test_emu_random_configurations(O, G, emu_lib, seed):
    T.test_random_configurations(O, G, emu_lib, seed)


@pytest.mark.parametrize("seed", range(0, 40, 4))
def test_emu_random_streams_all_decoder_paths(O, G, emu_lib, seed, monkeypatch):
    T.test_random_streams_all_decoder_paths(O, G, emu_lib, seed, monkeypatch)


# ---- the API-level GPU tests that need no device tensors (tests/test_gpu_api.py): reader robustness, options, metadata
def test_emu_error_paths_and_hostile_input(O, G, emu_lib):
    A.test_developer_settings_reach_the_coders_that_follow(O, G, emu_lib)
    A.test_error_paths(G, emu_lib)
    A.test_hostile_tables_and_index(O, G, emu_lib)
    A.test_damaged_streams_do_not_crash(O, G, emu_lib)


def test_emu_output_buffers_and_stats(O, G, emu_lib):
    A.test_output_buffer_ownership_and_pinned_option(O, G, emu_lib)


def test_emu_metadata_and_exif(O, G, emu_lib):
    A.test_encoder_metadata_orientation(O, G, emu_lib)
    A.test_encoder_custom_exif_tags(O, G, emu_lib)


def test_emu_marker_between_scans_like_the_reference(O, G, emu_lib, _ref_lib):
    """A baseline multi-scan file may legally carry DHT / DQT / COM segments between its scans. The reference reader does not accept them
    (src/gpujpeg_reader.c:1131-1145: any marker other than RSTn, EOI, SOS or APPn inside the scan data is an error), and the product mirrors
    that decision instead of guessing ("what the reader accepts"): the same return code from both."""
    import ctypes as C
    import numpy as np
    from conftest import natural_image, oracle_image
    case = ("ms", 96, 64, 1, 1, 75, 4, 0, None, 3)
    jpeg = O.encode(oracle_image(O, case), natural_image(96, 64, 3, seed=4))
    sos = [int(i) for i in np.nonzero((jpeg[:-1] == 0xFF) & (jpeg[1:] == 0xDA))[0]]
    assert len(sos) == 3
    com = np.array([0xFF, 0xFE, 0x00, 0x05, 0x61, 0x62, 0x63], np.uint8)  # COM "abc" in front of the second scan's SOS
    bad = np.concatenate([jpeg[:sos[1]], com, jpeg[sos[1]:]])

    def rc_of(lib):
        dec = G.Decoder(lib)
        out = G.DecoderOutput()
        out.type = G.DECODER_OUTPUT_INTERNAL_BUFFER
        b = np.ascontiguousarray(bad)
        rc = lib.L.gpujpeg_decoder_decode(dec.h, b.ctypes.data, b.size, C.byref(out))
        ok = np.array_equal(dec.decode(jpeg)[0], O.decode(jpeg)[0])  # the decoder still works
        dec.close()
        return rc, ok

    rc, ok = rc_of(emu_lib)
    rc_ref, _ = rc_of(_ref_lib)
    assert ok and rc == rc_ref and rc != 0, (rc, rc_ref)


@pytest.mark.parametrize("bc", T.BATCH_CASES, ids=[c[0] for c in T.BATCH_CASES])
def test_emu_frame_batches(O, G, emu_lib, bc, monkeypatch):
    T.test_frame_batches(O, G, emu_lib, bc, monkeypatch)


def test_emu_frame_batch_with_strangers(O, G, emu_lib):
    T.test_frame_batch_with_strangers(O, G, emu_lib)


def test_emu_frame_batch_argument_errors(O, G, emu_lib):
    T.test_frame_batch_argument_errors(O, G, emu_lib)


def test_emu_frame_batch_device_resident(O, G, emu_lib):
    T.test_frame_batch_device_resident(O, G, emu_lib)


def test_emu_frame_batch_separate_buffers(O, G, emu_lib):
    T.test_frame_batch_separate_buffers(O, G, emu_lib)
