#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) --kernel-trace --stats database as a
per-kernel table: calls, total ms, average us, share.  usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in rows:
        if pct < 0.005 and calls < 10:
            continue
        lines.append(f"| `{name[:110]}` | {calls} | {total / 1e3:.2f} | {avg:.2f} | {pct:.2f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(out)
    else:
        print(out)


if __name__ == "__main__":
    main()
