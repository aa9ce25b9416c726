"""Timeline of one pipeline from a rocprofv3 kernel trace: duration of every kernel of the path and the idle time in front of it (end of the
previous kernel -> its start), averaged per kernel name over the steady state.  usage: kernel_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
from collections import defaultdict
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if r["Kernel_Name"].startswith(("k_", "void k_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]  # steady state
dur, gap, n = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None:
        gap[name] += (s - prev_end) / 1e3
        dur[name] += (e - s) / 1e3
        n[name] += 1
    prev_end = e
tot = 0.0
for k in n:
    print(f"{k:42s} calls {n[k]:5d}  idle in front {gap[k] / n[k]:7.2f} us   duration {dur[k] / n[k]:7.2f} us")
    tot += (gap[k] + dur[k]) / n[k]
print(f"per frame (sum of the averages): {tot:.1f} us")
