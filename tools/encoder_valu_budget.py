"""Vector instructions of k_encode_rgb444 per phase (a per-phase SQ_INSTS_VALU budget against a written-down ideal).
Needs the trace build (`make -C gpujpeg_amd/csrc trace`). Run under the counter pass:

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d <dir> -- python tools/encoder_valu_budget.py --run
    python tools/encoder_valu_budget.py --report <dir>

--run encodes bench.py's 8K frame once per stop stamp (every wave ends at stamp n: gj_hip_trace_stop_encoder) and once in full, in that order;
--report reads the counter CSV, takes the k_encode_rgb444 dispatches in order and prints the differences per wave next to the ideal."""
import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS = ["pixels loaded + colour transform"]
for c in ("Y", "Cb", "Cr"):
    STAMPS += [f"{c}: fDCT + quantiser + zig-zag store", f"{c}: coder walk", f"{c}: bit positions (prefix sums)", f"{c}: merge into the window + drain"]
# what the arithmetic needs per wave of 64 block positions (one instruction = one wave64 vector instruction), written down before measuring:
#   colour: 64 pixels x (3 byte->float conversions + 4.5 packed FMAs/adds for the 3 x 3 matrix on pixel pairs + 3/4 of a pack to bytes) ~ 64 x 9 = 580,
#           + 24 loads' addressing and the edge masks ~ 60  -> 640
#   fDCT + quantiser: 64 byte->float, 2 x 8 one-dimensional AAN transforms on packed pairs (29 packed ops per 2 rows -> 8 x 29 = 232),
#           32 packed multiplies + 32 packed rounding adds, 48 16-bit stores' worth of packing (perm) ~ 30 -> ~ 420 per component
#   coder: 32 dword reads -> 64-bit non-zero mask ~ 80; per non-zero coefficient ~ 14 (ctz, clear, run, size, LUT address, value bits, shift in,
#           flush test); the 8K natural frame has 7.7 non-zero coefficients per luminance block and 1.6 per chrominance block at q75 -> 190 / 100;
#           positions ~ 40; merge + drain ~ 120  -> ~ 430 (Y), ~ 340 (Cb, Cr)
IDEAL = [640, 420, 270, 40, 120, 420, 180, 40, 120, 420, 180, 40, 120]
IDEAL_TAIL = 60  # tile size, group total, result words

ap = argparse.ArgumentParser()
ap.add_argument("--run", action="store_true")
ap.add_argument("--report", default=None)
ap.add_argument("--workload", default="8k")
args = ap.parse_args()

if args.run:
    import ctypes as C
    import torch
    import bench
    from gpujpeg_amd import libgpujpeg as G
    lib = G.Library(os.environ.get("GJ_TRACE_LIB") or os.path.join(ROOT, "gpujpeg_amd", "lib", "libgpujpeg_trace.so"))
    assert lib.L.gpujpeg_init_device(0, 0) == 0
    dev = torch.device("cuda", 0)
    spec = bench.Spec(lib, args.workload, "natural", 75, dev, 12345)
    L = bench.Lanes(lib, spec, dev, 1)
    L.set_stats(False)
    ln = L.lanes[0]
    lib.L.gj_hip_trace_stop_encoder.argtypes = [C.c_int]
    for stop in list(range(1, len(STAMPS) + 1)) + [1 << 30]:
        assert lib.L.gj_hip_trace_stop_encoder(stop) == 0
        try:
            L.encode(ln)
        except Exception:
            pass  # (a stopped launch leaves no valid stream: the call may report that)
        torch.cuda.synchronize()
    print("launched", len(STAMPS) + 1, "encodes")
    sys.exit(0)

rows = []
for f in glob.glob(os.path.join(args.report, "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
enc = {}
for r in rows:
    if "k_encode_rgb444" not in r["Kernel_Name"]:
        continue
    d = enc.setdefault(int(r["Dispatch_Id"]), {})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(enc)[-(len(STAMPS) + 1):]
vals = [enc[i]["SQ_INSTS_VALU"] for i in ids]
waves = enc[ids[-1]]["SQ_WAVES"]
print(f"k_encode_rgb444, {args.workload}: {int(waves)} waves, SQ_INSTS_VALU of the complete kernel {vals[-1] / 1e6:.2f} M = {vals[-1] / waves:.0f} per wave")
print(f"{'phase':48s} {'measured / wave':>16s} {'ideal':>8s}")
prev = 0.0
for name, v, ideal in zip(STAMPS, vals, IDEAL):
    print(f"{name:48s} {(v - prev) / waves:16.0f} {ideal:8d}")
    prev = v
print(f"{'tile size, group total (behind the last stamp)':48s} {(vals[-1] - prev) / waves:16.0f} {IDEAL_TAIL:8d}")
print(f"{'sum':48s} {vals[-1] / waves:16.0f} {sum(IDEAL) + IDEAL_TAIL:8d}")
