#!/usr/bin/env python
"""Print the per-kernel summary of a rocprofv3 `--kernel-trace --stats` run.

  python tools/top_kernels.py gpurun_out/<dir>/<name>_results.db [N] [--csv out.csv]

rocprofv3 (ROCm 7.2) writes an sqlite database by default; its `top_kernels` view holds
name / calls / total us / average us / percentage.
"""

from __future__ import annotations

import csv
import sqlite3
import sys


def main() -> None:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    db, n = args[0], int(args[1]) if len(args) > 1 else 30
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    print(f"total kernel time {total / 1e3:.2f} ms over {sum(r[1] for r in rows)} launches")
    for name, calls, dur, avg, pct in rows[:n]:
        print(f"{pct:6.2f}%  {dur / 1e3:9.3f} ms  {calls:6d} x {avg:9.2f} us  {name[:110]}")
    if "--csv" in sys.argv:
        out = sys.argv[sys.argv.index("--csv") + 1]
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
            w.writerows(rows)


if __name__ == "__main__":
    main()
