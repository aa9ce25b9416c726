#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats results database into the short per-kernel table kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({db_path.split('/')[-1]})\n")
        if note:
            f.write(f"# {note}\n")
        f.write("# durations in microseconds\n")
        f.write(f"{'kernel':<72} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}\n")
        for name, calls, total, avg, pct in rows:
            short = name.split("(")[0].replace("void ", "")
            if len(short) > 70:
                short = short[:67] + "..."
            f.write(f"{short:<72} {calls:>6} {total:>12.1f} {avg:>10.2f} {pct:>6.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
