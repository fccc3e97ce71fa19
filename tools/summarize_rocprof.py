#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV pair into a small markdown summary for profiles/.

usage: summarize_rocprof.py <dir-with-*_kernel_stats.csv> <out.md> [title]"""
import csv
import glob
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats"
    stats = glob.glob(os.path.join(d, "*kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    lines = [f"# {title}", "", "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r["Name"].replace("vits::", "").replace("void ", "")
        name = name.split("(")[0]
        lines.append(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
                     f"{int(r['MinNs']) / 1e3:.2f} | {int(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
