#!/usr/bin/env python3
"""Per-kernel LDS counters from tools/pmc_lds.sh.   usage: pmc_lds_table.py <dir> <out.md> "<profiled command>"
LDS active = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8 XCDs): the share of the dispatch's cycles in which a
CU's LDS was working on an index instruction; conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import glob
import os
import re
import sys


def short(n):
    return re.sub(r"\s+", "", n.replace("(anonymous namespace)::", "").replace("void ", "").replace("vits::", "").split("(")[0])


def main():
    d, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "lds*counter_collection.csv"))[0])):
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k, c in cnt.items():
        mean = lambda name: sum(c[name]) / max(1, len(c[name]))
        cyc = mean("GRBM_GUI_ACTIVE") / 8.0
        if not cyc or mean("SQ_INSTS_LDS") == 0:
            continue
        rows.append((cyc * len(c["GRBM_GUI_ACTIVE"]), k, len(c["GRBM_GUI_ACTIVE"]), cyc, mean("SQ_INSTS_LDS"), mean("SQ_LDS_IDX_ACTIVE"),
                     mean("SQ_LDS_BANK_CONFLICT"), mean("SQ_ACTIVE_INST_LDS"), mean("SQ_WAIT_INST_LDS"), mean("SQ_WAVE_CYCLES")))
    rows.sort(reverse=True)
    lines = [f"# {cmd}", "", __doc__.split("\n", 1)[1].strip(), "",
             "| kernel | launches | cycles / launch | LDS insts / launch | LDS active | conflict share | wave cycles waiting on LDS |",
             "|---|---|---|---|---|---|---|"]
    for _, k, n, cyc, insts, idx, conf, act, wait, wave in rows[:30]:
        lines.append(f"| `{k}` | {n} | {cyc:.0f} | {insts:.0f} | {idx / (256.0 * cyc):.3f} | {conf / idx if idx else 0:.3f} | {wait / wave if wave else 0:.3f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
