#!/usr/bin/env python3
"""Merge the per-workload <tag>_<workload>_traffic.json files of tools/profile.sh into one <tag>_traffic.json keyed by workload and stamped with
the hash of the device sources (bench.py quotes the figures only for the kernels they were measured on).
usage: merge_traffic.py <dir> <tag> <workload> ..."""
import json
import os
import sys

d, tag, names = sys.argv[1], sys.argv[2], sys.argv[3:]
out = {"note": "bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (separate rocprofv3 --pmc passes; the x 2 is the gfx950 correction of "
               "/opt/skills/guides/MI355X_MICROARCH.md); valu_insts = SQ_INSTS_VALU per launch (wave instructions); tools/profile.sh per workload",
       "workloads": {}}
hashes = set()
for n in names:
    p = os.path.join(d, f"{tag}_{n}_traffic.json")
    if not os.path.exists(p):
        continue
    j = json.load(open(p))
    hashes.add(j.get("source_hash"))
    out["workloads"][n] = {k: j[k] for k in ("kernels", "valu_insts", "waves") if k in j}
out["source_hash"] = hashes.pop() if len(hashes) == 1 else None
json.dump(out, open(os.path.join(d, f"{tag}_traffic.json"), "w"), indent=1, sort_keys=True)
print("merged", list(out["workloads"]), "source_hash", out["source_hash"])
