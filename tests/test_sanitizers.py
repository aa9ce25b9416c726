"""-m "not gpu": AddressSanitizer + UndefinedBehaviorSanitizer over the WHOLE library -- the host C (reader, writer, tables, image file
I/O) and, through the CPU execution model (tests/hipemu), every kernel -- driven by hostile input: the fuzzer of tools/fuzz_decoder.py
(damaged entropy data, stray / renumbered restart markers, truncation, damaged headers, over-subscribed DHT, lying APP13 index) in all
entropy decoder modes, the hostile-table corpus of the API tests, and truncated / damaged image files. A finding aborts the subprocess.
(the heap overflow of round 1 in gj_tables.c is the kind of bug this tier is for.)"""
import os
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU_DIR = os.path.join(HERE, "hipemu")
ASAN_DIR = os.path.join(EMU_DIR, "_build_asan")
ASAN_LIB = os.path.join(ASAN_DIR, "libgpujpeg_emu.so")
CLANG_RT = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"


@pytest.fixture(scope="session")
def asan_env():
    if not os.path.exists(CLANG_RT) or shutil.which("make") is None:
        pytest.skip("needs ROCm's clang with its AddressSanitizer runtime")
    import fcntl
    with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:  # (pytest-xdist: one worker builds, the others wait -- a relink under a running test removes its library)
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-s", "-j8", "-C", EMU_DIR, "SAN=1", "OPT=-O1", f"OUT={ASAN_DIR}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return dict(os.environ, LD_PRELOAD=CLANG_RT, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1",
                UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", GJ_FUZZ_LIB=ASAN_LIB, FUZZ_TRIALS="18")


def _run(env, args, timeout=900):
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True, errors="replace", timeout=timeout, env=env, cwd=ROOT)
    tail = (r.stdout[-1500:] + "\n" + "\n".join(ln for ln in r.stderr.splitlines() if not ln.startswith("[GPUJPEG]"))[-3000:])
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    return r.stdout


@pytest.mark.parametrize("config", ["rgb_auto", "rgb_r0", "rgb_il", "rgb_420_il", "uyvy_il", "gray", "rgba"])
def test_fuzzed_streams_under_sanitizers(asan_env, config):
    out = _run(asan_env, [os.path.join(ROOT, "tools", "fuzz_decoder.py"), config])
    assert "fuzz failures: 0" in out, out[-1500:]


def test_hostile_corpus_under_sanitizers(asan_env):
    _run(asan_env, [os.path.join(HERE, "sanitizer_driver.py"), ASAN_LIB])


def test_kernel_parity_tests_under_sanitizers(asan_env):
    """The parity tests of the CPU execution model (tests/test_emu_parity.py) once more with the instrumented build: the fuzzers above feed hostile
    STREAMS, these feed every geometry, pixel format, restart interval and developer switch to every kernel -- the two out-of-bounds accesses of
    round 4 (k_gather's preload behind the last, shorter tile of a frame; a forced marker-scan shape with more workgroups than the scratch had
    room for) were of this kind, silent on the GPU and an occasional crash of the plain build here."""
    env = dict(asan_env, GJ_EMU_LIB=ASAN_LIB)
    sel = "frame_batch or tiles_and_gather or marker_scan or token_mode or reuse_padding or segment_info or dense_two_bit or emu_encode_decode_bit_exact or random_configurations"
    out = _run(env, ["-m", "pytest", os.path.join(HERE, "test_emu_parity.py"), "-x", "-q", "-n", "4", "-p", "no:faulthandler", "-p", "no:cacheprovider", "-k", sel], timeout=1500)
    assert " passed" in out and "failed" not in out, out[-1500:]
