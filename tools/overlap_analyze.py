#!/usr/bin/env python3
"""How the front end and the generator share the GPU in throughput mode, from a rocprofv3 --kernel-trace CSV:
per queue the kernel count and busy time, and over the traced steps the time in which only front-end kernels, only
generator kernels, both or none were resident.  usage  python tools/overlap_analyze.py <kernel_trace.csv> [skip_ms]"""
import csv
import sys

GEN = ("@gen",)


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def front_end_block_time(rows):
    """What the front end takes from the generator beside it: a block of a front-end kernel holds one of the chip's
    block slots (two per CU for the generator's large kernels) for as long as its kernel runs -- an upper bound, in
    microseconds of the whole chip per launch = duration x min(blocks, 512) / 512, summed per kernel class."""
    gen_q = {r["Queue_Id"] for r in rows if "resblock_" in r["Kernel_Name"] or "conv_gl16" in r["Kernel_Name"]}
    agg = {}
    for r in rows:
        if r["Queue_Id"] in gen_q:
            continue
        name = r["Kernel_Name"].replace("void vits::", "").replace("(anonymous namespace)::", "").split("(")[0][:56]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        blocks = 1
        for ax in "XYZ":
            blocks *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
        a = agg.setdefault(name, [0, 0.0, 0.0, 0])
        a[0] += 1
        a[1] += d
        a[2] += d * min(blocks, 512) / 512
        a[3] += blocks
    total = sum(a[2] for a in agg.values())
    print(f"front-end kernels (every queue without a ResBlock kernel): {sum(a[0] for a in agg.values())} launches, "
          f"{total / 1e3:.2f} ms of chip-equivalent block time")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:12]:
        print(f"  {k:56s} {a[0]:5d} launches  avg {a[1] / a[0]:7.1f} us  avg {a[3] / a[0]:7.1f} blocks  {a[2] / 1e3:7.2f} ms chip-equivalent")


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
    ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows]
    ks.sort(key=lambda k: k[1])
    front_end_block_time(rows)
    # generator kernels: everything on a queue that carried a ResBlock kernel (the generator's stream and its side
    # streams; the front end has a queue of its own in throughput mode, and shares some kernel templates)
    gen_q = {k[3] for k in ks if "resblock_" in k[0] or "conv_gl16" in k[0]}
    ks = [(k[0] + (" @gen" if k[3] in gen_q else ""), k[1], k[2], k[3]) for k in ks]
    # the traced region: from the first generator kernel after `skip` to the last generator kernel
    gens = [k for k in ks if any(g in k[0] for g in GEN)]
    t_begin = gens[0][1] + skip
    t_end = gens[-1][2]
    ks = [k for k in ks if k[1] >= t_begin and k[2] <= t_end]
    gen = union([(k[1], k[2]) for k in ks if any(g in k[0] for g in GEN)])
    fe = union([(k[1], k[2]) for k in ks if not any(g in k[0] for g in GEN)])
    both = intersect(gen, fe)
    wall = t_end - t_begin
    print(f"window {wall / 1e6:.2f} ms, {len(ks)} kernels")
    print(f"generator kernels resident {length(gen) / 1e6:.2f} ms, front-end kernels resident {length(fe) / 1e6:.2f} ms, "
          f"both {length(both) / 1e6:.2f} ms, front end only {(length(fe) - length(both)) / 1e6:.2f} ms, "
          f"none {(wall - length(gen) - length(fe) + length(both)) / 1e6:.2f} ms")
    by_q = {}
    for k in ks:
        q = by_q.setdefault(k[3], [0, 0.0])
        q[0] += 1
        q[1] += (k[2] - k[1]) / 1e6
    for q, (n, ms) in sorted(by_q.items()):
        print(f"queue {q}: {n} kernels, {ms:.2f} ms of kernel time")
    # generator kernel durations while a front-end kernel is resident against the same kernel alone
    stat = {}
    for name, s, e, _ in ks:
        if not any(g in name for g in GEN):
            continue
        ov = length(intersect([[s, e]], fe)) / max(e - s, 1)
        key = name.split("(")[0].replace("void vits::", "").replace("(anonymous namespace)::", "")[:60]
        st = stat.setdefault(key, [[], []])
        st[0 if ov < 0.1 else 1].append((e - s) / 1e3)
    print("generator kernels: mean us without / with a front-end kernel resident (count)")
    for key, (a, b) in sorted(stat.items(), key=lambda kv: -sum(kv[1][0]) - sum(kv[1][1]))[:14]:
        ma = sum(a) / len(a) if a else float("nan")
        mb = sum(b) / len(b) if b else float("nan")
        print(f"  {key:60s} {ma:8.1f} ({len(a):3d})  {mb:8.1f} ({len(b):3d})")


if __name__ == "__main__":
    main()
