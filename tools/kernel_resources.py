#!/usr/bin/env python3
"""Compile one .hip file of the product for gfx950 and print registers / LDS / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py gpujpeg_amd/csrc/gj_decode.hip [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
       f"-I{ROOT}/include", f"-I{ROOT}/gpujpeg_amd/csrc", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for ln in out.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", ln) or re.search(r"Name: (\S+) \[", ln)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
        continue
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
if "error" in out and not rows:
    print(out)
print(f"{'kernel':70s} {'sgpr':>5s} {'vgpr':>5s} {'scr':>4s} {'lds':>6s} {'occ':>4s}")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:70]:70s} {r.get('sgpr',0):5d} {r.get('vgpr',0):5d} {r.get('scratch',0):4d} {r.get('lds',0):6d} {r.get('occ',0):4d}")
