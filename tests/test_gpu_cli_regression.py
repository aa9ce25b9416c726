"""The reference's CLI regression script (test/regression/run_tests.sh) replayed with our gpujpegtool on the GPU box: the same
command lines, the same pass criteria (PSNR floors of test/common.sh / run_tests.sh:116-151), with numpy in the place of
ImageMagick's compare -- and, where the script only checks that nothing crashes, the oracle's bytes as an extra check."""
import os
import subprocess

import numpy as np
import pytest

from conftest import psnr

pytestmark = pytest.mark.gpu


@pytest.fixture()
def tool(G, gpu_lib, tmp_path, monkeypatch):
    exe = os.path.join(os.path.dirname(G.PRODUCT_LIB), "gpujpegtool")
    monkeypatch.chdir(tmp_path)

    def run(*args, ok=True):
        r = subprocess.run([exe, *[str(a) for a in args]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=tmp_path, timeout=120)
        if ok:
            assert r.returncode == 0, (args, r.stderr.decode()[-400:])
        return r
    return run


def read_pnm(path):
    data = open(path, "rb").read()
    fields, pos = [], 0
    while len(fields) < 4:  # magic, width, height, maxval
        while data[pos:pos + 1].isspace():
            pos += 1
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        fields.append(data[pos:end])
        pos = end
    pos += 1
    w, h = int(fields[1]), int(fields[2])
    c = 3 if fields[0] == b"P6" else 1
    return np.frombuffer(data[pos:pos + w * h * c], np.uint8).reshape(h, w, c)


def read_pam(path):
    data = open(path, "rb").read()
    head, body = data.split(b"ENDHDR\n", 1)
    kv = dict(line.split(None, 1) for line in head.decode().splitlines()[1:] if " " in line)
    w, h, d = int(kv["WIDTH"]), int(kv["HEIGHT"]), int(kv["DEPTH"])
    return np.frombuffer(body[:w * h * d], np.uint8).reshape(h, w, d)


def test_commit_b620be2(tool):
    """All-zero input with restart interval 1 must decode to exactly zero (run_tests.sh:11-25)."""
    tool("-e", "-s", "1920x1080", "-r", "1", "-f", "444-u8-p0p1p2", "/dev/zero", "out.jpg")
    tool("-d", "out.jpg", "out.rgb")
    out = np.fromfile("out.rgb", np.uint8)
    assert out.size == 1920 * 1080 * 3 and psnr(out, np.zeros_like(out)) >= 50.0
    tool("-e", "-s", "16x16", "-r", "1", "-f", "u8", "/dev/zero", "out.jpg")
    tool("-d", "out.jpg", "out.r")
    out = np.fromfile("out.r", np.uint8)
    assert out.size == 256 and psnr(out, np.zeros_like(out)) >= 50.0


def test_different_sizes(tool):
    """Growing and shrinking images through one encoder and one decoder (commits e52abeab / 791a9e6b, run_tests.sh:28-48)."""
    enc, dec = [], []
    for dim in (32, 1024, 64, 2048):
        with open(f"{dim}.pnm", "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (dim, dim) + bytes(dim * dim * 3))
        enc += [f"{dim}.pnm", f"{dim}.jpg"]
        dec += [f"{dim}.jpg", f"{dim}.pam"]
    tool("-e", *enc)
    tool("-d", *dec)
    for dim in (32, 1024, 64, 2048):
        assert read_pam(f"{dim}.pam").shape == (dim, dim, 3)
        assert int(read_pam(f"{dim}.pam").max()) == 0


def test_decode_outside_pinned_and_huffman_buffer(tool, O):
    """392x386 grey noise (run_tests.sh:51-56): the decoded PNM holds the oracle's decoding of the JPEG we wrote."""
    tool("-e", "392x386.p_u8.noise.tst", "out.jpg")
    tool("-d", "out.jpg", "out.pnm")
    jpeg = np.fromfile("out.jpg", np.uint8)
    assert np.array_equal(read_pnm("out.pnm").reshape(-1), O.decode(jpeg)[0])


def test_postprocess_pitch_planar_422(tool):
    """1119x561 planar 4:2:2 to Y4M (run_tests.sh:58-62): odd width, planar output."""
    tool("-e", "1119x561.c_ycbcr-jpeg.p_422-u8-p0p1p2.tst", "ycbcr422.jpg")
    tool("-d", "-c", "ycbcr-jpeg", "ycbcr422.jpg", "out.y4m")
    head = open("out.y4m", "rb").readline()
    assert head.startswith(b"YUV4MPEG2 W1119 H561") and b"C422" in head


def test_nonexistent_input_fails(tool):
    assert tool("-e", "nonexistent.pam", "fail.jpg", ok=False).returncode != 0


def test_out_ext_XXX(tool):
    """RGBA with alpha kept, no colour transform, q99 (run_tests.sh:88-95): -b dumps the generated input as PAM, the decoder picks the
    PAM extension for .XXX, and the two files agree closely."""
    tool("-q", "99", "-e", "-b", "-Na", "1111x511.p_4444-u8-p0123.random.tst", "rgba.jpg")
    tool("-d", "-Na", "rgba.jpg", "test_out_ext_XXX.XXX")
    src, out = read_pam("input-1111x511.p_4444-u8-p0123.random.pam"), read_pam("test_out_ext_XXX.pam")
    assert src.shape == out.shape == (511, 1111, 4)
    assert psnr(out, src) >= 40.0


def test_pam_pnm_y4m_chain(tool):
    """run_tests.sh:98-111: every container on both ends."""
    w = h = 256
    with open("in.y4m", "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F25:1 Ip A0:0 C444 XCOLORRANGE=FULL\nFRAME\n" % (w, h) + bytes(w * h * 3))
    tool("-e", "in.y4m", "out.jpg")
    for ext in ("y4m", "pam", "pnm"):
        tool("-d", "out.jpg", f"out.{ext}")
    tool("-e", "out.pam", "out.jpg")
    tool("-e", "out.pnm", "out.jpg")
    assert os.path.getsize("out.jpg") > 600


@pytest.mark.parametrize("name,quality,flags,floor", [
    ("1119x561.random.c_rgb", 75, [], 22.0),
    ("1119x561.p_u8.random", 75, [], 28.4),
    ("1119x561.p_4444-u8-p0123.random", 90, ["-a", "-N"], 36.3),
])
def test_random_psnr(tool, name, quality, flags, floor):
    """The quality floors the reference holds itself to (run_tests.sh:116-151; its own values: 22.26 / 28.53 / 36.4 dB)."""
    tool(*flags, "-b", "-q", quality, "-e", "-s", "1119x561", f"{name}.tst", "in.jpg")
    tool(*flags, "-d", "in.jpg", f"out-{name}.XXX")
    src = [f for f in os.listdir(".") if f.startswith(f"input-{name}.")]
    out = [f for f in os.listdir(".") if f.startswith(f"out-{name}.")]
    assert len(src) == 1 and len(out) == 1 and src[0].rsplit(".", 1)[1] == out[0].rsplit(".", 1)[1]
    rd = read_pam if src[0].endswith(".pam") else read_pnm
    a, b = rd(src[0]), rd(out[0])
    assert a.shape == b.shape
    assert psnr(b, a) >= floor


def test_restart0_reconfiguration(tool, O):
    """-r 0 streams of two sizes, then both decoded by one decoder (run_tests.sh:153-161; the reference takes its CPU Huffman path)."""
    tool("-r", "0", "50x50.tst", "50.jpg")
    tool("-r", "0", "60x60.tst", "60.jpg")
    tool("-d", "60.jpg", "60.pnm", "50.jpg", "50.pnm")
    for n in (50, 60):
        jpeg = np.fromfile(f"{n}.jpg", np.uint8)
        assert np.array_equal(read_pnm(f"{n}.pnm").reshape(-1), O.decode(jpeg)[0])


COLOR_MODES = [
    # name, extension, the reference's MODE string (colors/run_tests.sh), bytes per frame as a function of (w, h)
    ("image_yuv_444p_subsampled", "yuv", "--colorspace=ycbcr-bt601 --subsampled --pixel-format=444-u8-p0p1p2"),
    ("image_yuv_422_interleaved", "yuv", "--pixel-format=422-u8-p1020 -i"),
    ("image_yuv_420p_native_709", "yuv", "-N --colorspace ycbcr-bt709 --pixel-format=420-u8-p0p1p2"),
    ("image_rgb_444", "rgb", "--colorspace=rgb --pixel-format=444-u8-p012"),
    ("image_rgb_444_native", "rgb", "-N --pixel-format=444-u8-p012"),
    ("image_rgb0_444_interleaved_subsampled", "rgba", "-i -S -f 4444-u8-p0123"),
]


@pytest.mark.parametrize("name,ext,mode", COLOR_MODES, ids=[m[0] for m in COLOR_MODES])
def test_colors_modes(tool, name, ext, mode):
    """colors/run_tests.sh + colors/test_common.sh: six pixel format / colour space / subsampling modes through the CLI at quality
    100 must come back with PSNR >= 40 dB. The reference feeds a camera picture converted by FFmpeg; here a smooth synthetic 1080p
    picture is laid out in each file format directly and compared with the decoded file in that same layout."""
    w, h = 1920, 1080
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    y = 128 + 80 * np.sin(xx / 97.0) * np.cos(yy / 61.0) + 20 * np.sin((xx + yy) / 23.0)
    cb = 128 + 50 * np.sin(xx / 211.0 + 1.0) * np.cos(yy / 173.0)
    cr = 128 + 50 * np.cos(xx / 157.0) * np.sin(yy / 199.0 + 0.5)
    q = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)  # noqa: E731
    if "444-u8-p0p1p2" in mode:
        raw = np.concatenate([q(y).ravel(), q(cb).ravel(), q(cr).ravel()])
    elif "422-u8-p1020" in mode:
        pk = np.empty((h, w, 2), np.uint8)
        pk[:, :, 1] = q(y)
        pk[:, 0::2, 0] = q(cb)[:, 0::2]
        pk[:, 1::2, 0] = q(cr)[:, 0::2]
        raw = pk.ravel()
    elif "420-u8-p0p1p2" in mode:
        raw = np.concatenate([q(y).ravel(), q(cb)[0::2, 0::2].ravel(), q(cr)[0::2, 0::2].ravel()])
    elif ext == "rgba":
        raw = np.stack([q(y), q(cb), q(cr), np.full((h, w), 255, np.uint8)], -1).ravel()
    else:
        raw = np.stack([q(y), q(cb), q(cr)], -1).ravel()
    raw.tofile(f"{name}.{ext}")
    tool("--size", "1920x1080", *mode.split(), "--encode", "--quality", "100", f"{name}.{ext}", f"{name}.encoded.jpg")
    tool(*mode.split(), "--decode", f"{name}.encoded.jpg", f"{name}.decoded.{ext}")
    back = np.fromfile(f"{name}.decoded.{ext}", np.uint8)
    if ext == "rgba":  # decoded without -a: the alpha channel is dropped (main.c:268-271), compare the colour channels
        assert back.size in (w * h * 3, w * h * 4)
        a = raw.reshape(h, w, 4)[..., :3]
        b = back.reshape(h, w, -1)[..., :3]
        assert psnr(b, a) >= 40.0
    else:
        assert back.size == raw.size

        def to_rgb(buf):  # what the reference's script does with FFmpeg before comparing: everything to rgb24 (BT.601, limited range)
            if ext == "rgb":
                return buf.astype(np.float32)
            if "444-u8-p0p1p2" in mode:
                yv, u, v = buf.reshape(3, h, w).astype(np.float32)
            elif "422-u8-p1020" in mode:
                pk = buf.reshape(h, w, 2).astype(np.float32)
                yv, u, v = pk[:, :, 1], np.repeat(pk[:, 0::2, 0], 2, 1), np.repeat(pk[:, 1::2, 0], 2, 1)
            else:
                yv = buf[:w * h].reshape(h, w).astype(np.float32)
                u = np.repeat(np.repeat(buf[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), 2, 0), 2, 1).astype(np.float32)
                v = np.repeat(np.repeat(buf[w * h * 5 // 4:].reshape(h // 2, w // 2), 2, 0), 2, 1).astype(np.float32)
            yv, u, v = 1.164 * (yv - 16), u - 128, v - 128
            return np.clip(np.stack([yv + 1.596 * v, yv - 0.392 * u - 0.813 * v, yv + 2.017 * u], -1), 0, 255)
        assert psnr(to_rgb(back), to_rgb(raw)) >= 40.0
