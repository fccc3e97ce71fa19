#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats result into a small markdown summary for profiles/.

usage: summarize_rocprof.py <dir-with-*_kernel_stats.csv | results.db> <out.md> [title]
The input is either the directory of a `--output-format csv` run or the rocpd sqlite database of a default run."""
import csv
import glob
import os
import sqlite3
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("vits::", "").replace("void ", "").split("(")[0]


def rows_from_csv(d):
    stats = glob.glob(os.path.join(d, "*kernel_stats.csv"))[0]
    for r in csv.DictReader(open(stats)):
        yield (short(r["Name"]), int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), int(r["MinNs"]),
               int(r["MaxNs"]), float(r["Percentage"]))


def rows_from_db(path):
    con = sqlite3.connect(path)
    agg = {}
    for name, start, end in con.execute("select name, start, end from kernels"):
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        dur = end - start
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        yield name, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / total


def main():
    src, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats"
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    lines = [f"# {title}", "", "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx, pct in rows:
        lines.append(f"| `{name}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {pct:.2f} |")
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
