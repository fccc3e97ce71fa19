#!/usr/bin/env python3
"""Print the kernels of the last infer call in a rocprofv3 --kernel-trace database, in launch order.

usage: trace_sequence.py <results.db> [max_rows]"""
import sqlite3
import sys


def short(n):
    return n.replace("vits::", "").replace("void ", "").split("(")[0]


def main():
    con = sqlite3.connect(sys.argv[1])
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rows = list(con.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
    first = [i for i, r in enumerate(rows) if "cond_gemv_kernel" in r[0]][-1]
    seq = rows[first:first + limit]
    t0 = seq[0][1]
    prev_end = t0
    for name, s, e, gx, gy, gz, wx in seq:
        print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:8.1f}  {short(name):38s} "
              f"grid {gx // max(wx, 1)}x{gy}x{gz}")
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
