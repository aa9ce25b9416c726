"""Largest time gaps in an AMD_LOG_LEVEL log (lines carry '<n> us:'): prints the lines around each gap above a threshold."""
import re, sys
path, thr_us, ctx = sys.argv[1], float(sys.argv[2]) * 1e3, int(sys.argv[3]) if len(sys.argv) > 3 else 25
pat = re.compile(r"(\d+) us:")
lines, ts = [], []
with open(path, errors="replace") as f:
    for ln in f:
        m = pat.search(ln)
        if m:
            lines.append(ln.rstrip()[:260]); ts.append(int(m.group(1)))
print("lines", len(lines))
order = sorted(range(len(ts)), key=lambda i: ts[i])   # several threads: sort by time
ts2 = [ts[i] for i in order]
gaps = [(ts2[k + 1] - ts2[k], k) for k in range(5000, len(ts2) - 2000) if ts2[k + 1] - ts2[k] > thr_us]  # (start-up and tear-down are not the question)
print("gaps above threshold:", [(g, k) for g, k in gaps][:10])
for g, k in gaps[:3]:
    print(f"==== gap {g / 1e3:.1f} ms after sorted line {k}")
    for q in range(max(0, k - ctx), min(len(order), k + ctx)):
        print(("-> " if q == k + 1 else "   ") + lines[order[q]])
