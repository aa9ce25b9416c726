"""The C-ABI library loads on a machine without a GPU and exports every symbol that include/*.h declares;
struct layouts seen through ctypes equal what a C compiler sees through the headers. No compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def declared_functions():
    names = set()
    for d, _, files in os.walk(INC):
        for f in files:
            if not f.endswith(".h"):
                continue
            text = open(os.path.join(d, f)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            for m in re.finditer(r"(?:GPUJPEG_API|GJ_HIP_API)\b[^;{]*?\b((?:gpujpeg|gj_hip)_[a-z0-9_]+)\s*\(", text, flags=re.S):
                names.add(m.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) > 100
    missing = [n for n in names if not hasattr(lib.L, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"


def test_reference_api_coverage(lib):
    """Every function of the reference's public headers exists with the same name (drop-in)."""
    ref_inc = "/root/reference/libgpujpeg"
    if not os.path.isdir(ref_inc):
        pytest.skip("reference headers not available here")
    names = set()
    for f in os.listdir(ref_inc):
        if f.endswith(".h"):
            names |= set(re.findall(r"\b(gpujpeg_[a-z0-9_]+)\s*\(", open(os.path.join(ref_inc, f)).read()))
    missing = sorted(n for n in names if not hasattr(lib.L, n))
    assert not missing, missing


C_PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "libgpujpeg/gpujpeg.h"
#define P(t) printf(#t " %zu\n", sizeof(t))
#define O(t, m) printf(#t "." #m " %zu\n", offsetof(t, m))
int main(void) {
    P(struct gpujpeg_parameters); P(struct gpujpeg_image_parameters); P(struct gpujpeg_encoder_input);
    P(struct gpujpeg_decoder_output); P(struct gpujpeg_image_info); P(struct gpujpeg_duration_stats);
    P(struct gpujpeg_decoder_init_parameters); P(struct gpujpeg_devices_info); P(struct gpujpeg_opengl_texture);
    O(struct gpujpeg_parameters, sampling_factor); O(struct gpujpeg_parameters, color_space_internal);
    O(struct gpujpeg_decoder_output, param_image); O(struct gpujpeg_decoder_output, metadata);
    printf("RESTART_AUTO %d\nGPUJPEG_PIXFMT_AUTODETECT %d\nGPUJPEG_SUBSAMPLING_420 %u\nGPUJPEG_IMAGE_FILE_TST %d\n",
           RESTART_AUTO, (int)GPUJPEG_PIXFMT_AUTODETECT, GPUJPEG_SUBSAMPLING_420, (int)GPUJPEG_IMAGE_FILE_TST);
    return 0;
}
"""


def probe(include_dir, tmp_path, tag, extra=()):
    src = tmp_path / f"probe_{tag}.c"
    exe = tmp_path / f"probe_{tag}"
    src.write_text(C_PROBE)
    subprocess.check_call(["gcc", "-std=c11", "-I", include_dir, *extra, str(src), "-o", str(exe)])
    return dict(l.rsplit(" ", 1) for l in subprocess.check_output([str(exe)], text=True).strip().splitlines())


def test_struct_layouts_match_ctypes_and_reference(G, tmp_path):
    ours = probe(INC, tmp_path, "ours")
    assert int(ours["struct gpujpeg_parameters"]) == C.sizeof(G.Parameters)
    assert int(ours["struct gpujpeg_image_parameters"]) == C.sizeof(G.ImageParameters)
    assert int(ours["struct gpujpeg_encoder_input"]) == C.sizeof(G.EncoderInput)
    assert int(ours["struct gpujpeg_decoder_output"]) == C.sizeof(G.DecoderOutput)
    assert int(ours["struct gpujpeg_image_info"]) == 512 == C.sizeof(G.ImageInfo)
    assert int(ours["struct gpujpeg_duration_stats"]) == C.sizeof(G.DurationStats)
    if os.path.isdir("/root/reference/libgpujpeg"):
        gen = tmp_path / "gen" / "libgpujpeg"
        gen.mkdir(parents=True)
        (gen / "gpujpeg_version.h").write_text("#define GPUJPEG_VERSION_MAJOR 0\n#define GPUJPEG_VERSION_MINOR 27\n#define GPUJPEG_VERSION_PATCH 13\n")
        theirs = probe("/root/reference", tmp_path, "ref", extra=["-I", str(tmp_path / "gen")])
        assert ours == theirs, "public struct layout / constant drift versus the reference headers"


def test_developer_settings_api_and_no_environment(lib, G):
    """(round 6) The release library does not read the environment: it imports no getenv, and the developer settings go through
    gpujpeg_amd_tuning (include/gpujpeg_amd_ext.h) -- known names accepted with or without a value, unknown names refused, NULL forgets everything."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", G.PRODUCT_LIB], capture_output=True, text=True).stdout
    assert "getenv" not in syms, "the release library must not read the environment"
    names = lib.tuning_names()
    assert {"GJ_DEC_TOKENS", "GJ_DEC_NO_TOKENS", "GJ_ENC_TAIL", "GJ_DEC_SEQ", "GJ_COPY_LANES", "GPUJPEG_NO_FUSED"} <= set(names)
    for n in names:
        assert lib.tuning(n + "=1") or n == "", n
    assert lib.tuning("GJ_DEC_NO_SPEC")          # a name alone
    assert not lib.tuning("GJ_NO_SUCH_SETTING=1")  # unknown
    assert not lib.tuning("=1") and not lib.tuning("")
    assert lib.tuning(None)                      # back to the defaults
    # the helper the tests and tools use hands over exactly what an environment names
    G.apply_environment_settings(lib, {"GJ_DEC_TOKENS": "1", "HOME": "/nowhere"})
    assert lib.tuning(None)

