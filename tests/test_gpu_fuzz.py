"""-m gpu: the decoder fuzzer (tools/fuzz_decoder.py) on the real device with a bounded trial count: damaged streams in all entropy decoder
modes (sub-sequence kernels, token modes, lane-per-segment kernel) must end in an error return or an image, never in a GPU fault or a hang,
and the decoder must work afterwards. (The same fuzzer runs under AddressSanitizer on the CPU execution model: tests/test_sanitizers.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config", ["rgb_auto", "rgb_il", "uyvy_il", "rgb_420_il"])
def test_fuzzed_streams_on_the_device(gpu_lib, config):
    env = dict(os.environ, FUZZ_TRIALS="18")
    env.pop("GJ_FUZZ_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_decoder.py"), config], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "fuzz failures: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
