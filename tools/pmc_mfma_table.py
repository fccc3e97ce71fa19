#!/usr/bin/env python3
"""Per-kernel matrix-pipe occupancy and shader clock from tools/pmc_mfma.sh's two runs.

usage: pmc_mfma_table.py <dir> <out.md> "<profiled command>" [traffic.json]
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); durations come from the kernel trace of the
same command (the counter pass serialises and slows the launches).  clock = GRBM_GUI_ACTIVE / 8 XCDs / duration is
printed only for dispatches of 0.3 ms and more: on shorter ones the quotient reads high (ramp and drain are a large
share of the dispatch), and the clock a kernel holds is better taken from in-kernel stamps (tools/stamp_clock.py)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(n):
    return re.sub(r"\s+", "", n.replace("(anonymous namespace)::", "").replace("void ", "").replace("vits::", "").split("(")[0])


def main():
    d, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    traffic = json.load(open(sys.argv[4]))["kernels"] if len(sys.argv) > 4 else {}
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "mf*counter_collection.csv"))[0])):
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":           # the pass's own duration of this dispatch
            cnt[short(r["Kernel_Name"])]["pass_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "kt*kernel_trace.csv"))[0])):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for k, c in cnt.items():
        if k not in dur or "GRBM_GUI_ACTIVE" not in c:
            continue
        mean = lambda name: sum(c[name]) / len(c[name])
        us = sum(dur[k]) / len(dur[k]) / 1e3
        cyc = mean("GRBM_GUI_ACTIVE") / 8.0
        busy = mean("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc) if cyc else 0.0
        clock = cyc / mean("pass_ns")                        # GHz: cycles and nanoseconds of the same dispatches
        rows.append((sum(dur[k]) / 1e6, k, len(dur[k]), us, busy, mean("SQ_INSTS_MFMA"), clock))
    rows.sort(reverse=True)
    lines = [f"# {cmd}", "",
             "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) and clock = GRBM_GUI_ACTIVE / 8 / "
             "the dispatch's own duration, all from one `--pmc` pass; avg us from a `--kernel-trace` run of the same "
             "command; raw TFLOP/s = MFMA instructions x 32768 FLOP (32x32x16; 16384 for conv_gl16_kernel's 16x16x32) / avg us; "
             "clock only for dispatches of 0.3 ms and more (it reads high on shorter ones); HBM GB/s = (2 x FETCH_SIZE + "
             "WRITE_SIZE) / avg us where the traffic table has the kernel.",
             "", "| kernel | launches | total ms | avg us | MFMA busy | clock GHz | busy x clock | MFMA insts / launch | raw TFLOP/s | HBM GB/s |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for tot, k, n, us, busy, insts, clock in rows[:36]:
        flop = 16384 if k.startswith("conv_gl16_kernel") else 32768      # v_mfma 16x16x32 against 32x32x16 (f16 / bf16)
        long_enough = us >= 300.0
        key = re.sub(r",true,(\d+)>", lambda m: ",staged>" if m.group(1) == "0" else f",staged,{m.group(1)}>", k)
        key = key.replace(",false,0>", ">").replace(",true>", ",staged>").replace(",false>", ">")
        hb = traffic.get(key, traffic.get(k, {})).get("hbm_bytes")
        gbps = f"{hb / us / 1e3:.0f}" if hb else ""
        lines.append(f"| `{k}` | {n} | {tot:.3f} | {us:.1f} | {busy:.3f} | {f'{clock:.2f}' if long_enough else ''} | "
                     f"{f'{busy * clock:.2f}' if long_enough else ''} | {insts:.0f} | {insts * flop / us / 1e6:.0f} | {gbps} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:20]))


if __name__ == "__main__":
    main()
