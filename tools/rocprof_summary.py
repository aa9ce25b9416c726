#!/usr/bin/env python3
"""Condense the rocprofv3 CSV output of tools/profile.sh into the short tables kept under profiles/.

usage: rocprof_summary.py <dir with prof_stats/ prof_fetch/ prof_write/> <tag>
writes <dir>/<tag>_kernel_stats.txt and <dir>/<tag>_hbm_traffic.txt (copy them to profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, suffix):
    f = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return f[0] if f else None


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n if len(n) <= 70 else n[:67] + "..."


def kernel_stats(d, out, note):
    f = find(os.path.join(d, "prof_stats"), "kernel_stats.csv")
    if not f:
        print("no kernel_stats.csv under", d)
        return
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats summary\n# " + note + "\n# durations in microseconds\n")
        o.write(f"{'kernel':<72} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}\n")
        for r in rows:
            o.write(f"{short(r['Name']):<72} {int(r['Calls']):>6} {float(r['TotalDurationNs']) / 1e3:>12.1f} "
                    f"{float(r['AverageNs']) / 1e3:>10.2f} {float(r['Percentage']):>6.2f}\n")
    print(open(out).read())


def counter_table(d, sub, counter):
    f = find(os.path.join(d, sub), "counter_collection.csv")
    acc = defaultdict(lambda: [0, 0.0])
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def traffic(d, out, note):
    fe = counter_table(d, "prof_fetch", "FETCH_SIZE")
    wr = counter_table(d, "prof_write", "WRITE_SIZE")
    with open(out, "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per-dispatch averages, raw counter values in KiB\n")
        o.write("# " + note + "\n")
        o.write(f"{'kernel':<72} {'disp':>5} {'FETCH_SIZE_avg':>15} {'WRITE_SIZE_avg':>15}\n")
        for k in sorted(set(fe) | set(wr), key=lambda k: -(fe[k][1] + wr[k][1])):
            nf, sf = fe[k]
            nw, sw = wr[k]
            o.write(f"{k:<72} {max(nf, nw):>5} {sf / nf if nf else 0:>15.1f} {sw / nw if nw else 0:>15.1f}\n")
    print(open(out).read())


ENCODER_KERNELS = ("k_encode_", "k_fused_", "k_preprocess", "k_copy_planes_in", "k_dct", "k_huffman<", "k_huffman(", "k_scan_segments", "k_assemble", "k_gather", "k_segment_info")


def traffic_json(d, out, note):
    """HBM bytes per launch for bench.py's roofline.traffic: FETCH_SIZE (KiB, 64 B units counted per 128 B request on gfx950 -> x 2,
    /opt/skills/guides/MI355X_MICROARCH.md) + WRITE_SIZE (KiB)"""
    import json
    fe = counter_table(d, "prof_fetch", "FETCH_SIZE")
    wr = counter_table(d, "prof_write", "WRITE_SIZE")
    kernels = {}
    for k in set(fe) | set(wr):
        if not k.startswith("k_"):
            continue
        nf, sf = fe[k]
        nw, sw = wr[k]
        base = k.split("<")[0]
        name = ("enc:" if (k + "(").startswith(ENCODER_KERNELS) or base.startswith(ENCODER_KERNELS) else "dec:") + base
        kernels[name] = int(((sf / nf if nf else 0) * 2 + (sw / nw if nw else 0)) * 1024)
    # vector instructions per launch (SQ_INSTS_VALU, own PMC pass): what bounds these kernels is the issue rate, not HBM
    va = counter_table(d, "prof_sq", "SQ_INSTS_VALU")
    wv = counter_table(d, "prof_sq", "SQ_WAVES")
    valu, waves = {}, {}
    for k in va:
        if not k.startswith("k_") or not va[k][0]:
            continue
        base = k.split("<")[0]
        name = ("enc:" if (k + "(").startswith(ENCODER_KERNELS) or base.startswith(ENCODER_KERNELS) else "dec:") + base
        valu[name] = int(va[k][1] / va[k][0])
        waves[name] = int(wv[k][1] / wv[k][0]) if wv[k][0] else 0
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from gpujpeg_amd.source_hash import kernel_source_hash
        src_hash = kernel_source_hash()
    except Exception:
        src_hash = None
    json.dump({"note": note + "; bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (separate --pmc passes); valu_insts = SQ_INSTS_VALU per launch (wave instructions)",
               "source_hash": src_hash, "kernels": kernels, "valu_insts": valu, "waves": waves}, open(out, "w"), indent=1, sort_keys=True)
    with open(out.replace("_traffic.json", "_sq_counters.txt"), "w") as o:
        o.write("# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY (one pass), per-dispatch averages\n# " + note + "\n")
        names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"]
        tabs = {n: counter_table(d, "prof_sq", n) for n in names}
        for k in sorted(va, key=lambda k: -va[k][1]):
            if k.startswith("k_"):
                o.write(f"{k:<60} " + " ".join(f"{n}={tabs[n][k][1] / max(tabs[n][k][0], 1):.4g}" for n in names) +
                        f" VALU_per_wave={va[k][1] / max(wv[k][1], 1):.0f}\n")
    print(open(out.replace("_traffic.json", "_sq_counters.txt")).read())


if __name__ == "__main__":
    d, tag = sys.argv[1], sys.argv[2]
    note = " ".join(sys.argv[3:]) or "cmd: see tools/profile.sh (python bench.py, 8K RGB q75 natural pattern)"
    kernel_stats(d, os.path.join(d, tag + "_kernel_stats.txt"), note)
    traffic(d, os.path.join(d, tag + "_hbm_traffic.txt"), note)
    traffic_json(d, os.path.join(d, tag + "_traffic.json"), note)
