"""Decoder robustness sweep: damaged streams must end in an error return or a decoded image, never in a fault or a hang, and the decoder
object must work afterwards. One subprocess per (configuration, mode) so that a fault is attributed; the trial number is printed before
each decode. On a GPU box it drives the product; with GJ_FUZZ_LIB=<path> any other build of the library -- tests/test_sanitizers.py runs it
on the AddressSanitizer + UndefinedBehaviorSanitizer build of the CPU execution model (tests/hipemu), tests/test_gpu_fuzz.py on the GPU."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # name: (w, h, pixel format, colour space, quality, restart, interleaved, subsampling, output (pf, cs) or None)
    "rgb_auto": (640, 368, 1, 1, 75, -1, 0, None, None),
    "rgb_r0": (256, 192, 1, 1, 80, 0, 0, None, None),
    "rgb_il": (333, 123, 1, 1, 75, 5, 1, None, None),
    "rgb_420_il": (322, 242, 1, 1, 75, 3, 1, [(2, 2), (1, 1), (1, 1)], None),
    "uyvy_il": (640, 80, 3, 3, 90, 6, 1, None, (3, 3)),
    "gray": (333, 111, 0, 3, 75, -1, 0, None, None),
    "rgba": (200, 100, 6, 1, 75, 4, 1, [(1, 1)] * 4, None),
}
TRIALS = int(os.environ.get("FUZZ_TRIALS", "24"))


def child(name, mode):
    import oracle as O
    from conftest import natural_image, oracle_image
    from gpujpeg_amd import libgpujpeg as G
    lib = G.Library(os.environ.get("GJ_FUZZ_LIB") or None)
    G.apply_environment_settings(lib)  # (whatever the caller's environment names)
    if mode in ("tokens", "batchtok"):
        lib.tuning("GJ_DEC_TOKENS=1")
    if mode == "seq":  # the lane-per-segment entropy decoder over an LDS stage (plane mode)
        lib.tuning("GJ_DEC_NO_TOKENS")
        lib.tuning("GJ_DEC_SEQ=1")
    if mode == "seqtok":  # the same kernel in token mode
        lib.tuning("GJ_DEC_TOKENS=1")
        lib.tuning("GJ_DEC_SEQ=1")
    assert lib.L.gpujpeg_init_device(0, 0) == 0
    w, h, pf, cs, q, ri, il, ss, outfmt = CONFIGS[name]
    case = (name, w, h, pf, cs, q, ri, il, ss, 3)
    comps = {0: 1, 1: 3, 6: 4}.get(pf)
    raw = natural_image(w, h, comps, seed=w) if comps else O.noise(O.raw_size(w, h, pf), seed=w)
    jpeg = O.encode(oracle_image(O, case), raw)
    want = (O.decode(jpeg, *outfmt) if outfmt else O.decode(jpeg))[0]
    dec = G.Decoder(lib)
    if outfmt:
        dec.set_output_format(outfmt[1], outfmt[0])
    assert np.array_equal(dec.decode(jpeg)[0], want)
    hdr = int(np.nonzero((jpeg[:-1] == 0xFF) & (jpeg[1:] == 0xDA))[0][0]) + 14  # first byte of entropy-coded data (roughly)
    outcomes = {"decoded": 0, "error": 0}
    for t in range(TRIALS):
        rng = np.random.default_rng(1000 + t)
        bad = jpeg.copy()
        kind = t % 9
        if kind == 0:    # byte flips in the entropy-coded data
            idx = rng.integers(hdr, bad.size - 2, size=1 + t)
            bad[idx] = rng.integers(0, 256, size=idx.size, dtype=np.uint8)
        elif kind == 1:  # stray restart markers
            for i in rng.integers(hdr, bad.size - 4, size=1 + t // 6):
                bad[i], bad[i + 1] = 0xFF, 0xD0 + int(rng.integers(0, 8))
        elif kind == 2:  # truncation
            bad = bad[: int(bad.size * rng.uniform(0.1, 0.98))].copy()
        elif kind == 3:  # damage in the headers (tables, frame header)
            idx = rng.integers(20, hdr, size=2)
            bad[idx] = rng.integers(0, 256, size=idx.size, dtype=np.uint8)
        elif kind == 4:  # zeros / 0xFF runs
            a = int(rng.integers(hdr, bad.size - 64))
            bad[a:a + int(rng.integers(1, 64))] = 0xFF if t & 8 else 0
        elif kind == 5:  # deleted bytes (everything after shifts)
            a = int(rng.integers(hdr, bad.size - 8))
            bad = np.concatenate([bad[:a], bad[a + int(rng.integers(1, 5)):]])
        elif kind == 6:  # over-subscribed Huffman table: the code counts of the first DHT replaced
            d = int(np.nonzero((bad[:-1] == 0xFF) & (bad[1:] == 0xC4))[0][0])
            bad[d + 5:d + 21] = rng.integers(0, 256, size=16, dtype=np.uint8) if t & 8 else np.array([255] + [0] * 15, np.uint8)
        elif kind == 8:  # restart markers renumbered, doubled or removed (the reader's resynchronisation, src/gpujpeg_reader.c:1074-1108)
            r = np.nonzero((bad[:-1] == 0xFF) & (bad[1:] >= 0xD0) & (bad[1:] <= 0xD7))[0]
            if r.size:
                for i in rng.choice(r, size=min(r.size, 1 + t // 9), replace=False):
                    op = int(rng.integers(0, 3))
                    if op == 0:
                        bad[i + 1] = 0xD0 + int(rng.integers(0, 8))
                    elif op == 1:
                        bad[i], bad[i + 1] = 0x12, 0x34
                    else:
                        bad[i + 2:i + 4] = bad[i:i + 2]
        else:            # APP13 segment index with arbitrary offsets in front of the first scan
            sos = int(np.nonzero((bad[:-1] == 0xFF) & (bad[1:] == 0xDA))[0][0])
            n = int(rng.integers(1, 64))
            body = rng.integers(0, 256, size=4 * n, dtype=np.uint8) if t & 8 else np.sort(rng.integers(0, bad.size, size=n)).astype(">u4").view(np.uint8)
            app13 = np.concatenate([np.array([0xFF, 0xED, (3 + body.size) >> 8, (3 + body.size) & 255, 0], np.uint8), body])
            bad = np.concatenate([bad[:sos], app13, bad[sos:]])
        print(f"trial {t} kind {kind}", flush=True)
        if mode.startswith("batch"):
            # the damaged stream between two good ones in ONE batch call (gpujpeg_amd_decoder_decode_batch: every kernel launched once for the three):
            # the call fails as a whole or decodes, and a decoded good frame is exactly what it is alone -- nothing leaks between the frames of a batch
            try:
                px, _ = dec.decode_batch([jpeg, bad, jpeg])
                assert np.array_equal(px[0], want) and np.array_equal(px[2], want), f"a good frame of the batch differs after trial {t}"
                outcomes["decoded"] += 1
            except RuntimeError:
                outcomes["error"] += 1
            assert np.array_equal(dec.decode(jpeg)[0], want), f"decoder damaged after trial {t}"
            continue
        try:
            dec.decode(bad)
            outcomes["decoded"] += 1
        except Exception:
            outcomes["error"] += 1
        assert np.array_equal(dec.decode(jpeg)[0], want), f"decoder damaged after trial {t}"
    print("DONE", outcomes, flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3:
        child(sys.argv[1], sys.argv[2])
        sys.exit(0)
    only = [a for a in sys.argv[1:] if a in CONFIGS]
    # (the batch calls cover every layout: plane mode everywhere, token mode for the non-interleaved 4:4:4 configuration)
    jobs = [(name, mode) for name in (only or CONFIGS) for mode in ("default", "tokens", "seq", "seqtok", "batch") + (("batchtok",) if name == "rgb_auto" else ())]

    def one(job):
        name, mode = job
        try:
            r = subprocess.run([sys.executable, __file__, name, mode], capture_output=True, text=True, timeout=600)
            lines = r.stdout.strip().splitlines()
            ok = r.returncode == 0 and lines and lines[-1].startswith("DONE")
            return ok, " ".join(str(x) for x in (name, mode, "rc", r.returncode, lines[-1] if lines else "", "" if ok else "| " + (r.stderr.strip().splitlines() or [""])[-1][:200]))
        except subprocess.TimeoutExpired:
            return False, f"{name} {mode} TIMEOUT"

    # the children are independent processes (a decoder each): a few at a time (FUZZ_JOBS, default 3; on the GPU they share the device)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, int(os.environ.get("FUZZ_JOBS", "3")))) as pool:
        results = list(pool.map(one, jobs))
    for ok, line in results:
        print(line, flush=True)
    bad = sum(0 if ok else 1 for ok, _ in results)
    print("fuzz failures:", bad)
    sys.exit(1 if bad else 0)
