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


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
    ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows]
    ks.sort(key=lambda k: k[1])
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
