#!/usr/bin/env python3
"""The figures of a bench.py line a session log wants at a glance. usage: bench_brief.py <file with the JSON line>"""
import json
import sys

try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    sys.exit(f"no bench line: {e}")
r = d.get("roofline", {})
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "| dominant", r.get("kernel"), r.get("ms"), "frac", r.get("frac"))
print("frac_path", r.get("frac_path"), "enc_dir", r.get("frac_encode_direction"), "dec_dir", r.get("frac_decode_direction"),
      "hbm_read_enc", r.get("frac_hbm_read_encode"), "event_gap", r.get("event_gap_ms"))
print({k["kernel"]: k["ms"] for k in r.get("by_kernel", [])})
for k in ("encode_only", "decode_only"):
    if k in d:
        print(k, d[k].get("mpix_s"))
for k, v in d.get("workloads", {}).items():
    if isinstance(v, dict) and "mpix_s" in v:
        print(" ", k, v["mpix_s"], v.get("solo_gpu_ms"), "enc/dec only", v.get("encode_only_mpix_s"), v.get("decode_only_mpix_s"))
