#!/usr/bin/env python3
"""Turn the two counter passes of tools/pmc_traffic.sh into the per-kernel HBM-traffic table bench.py reads.

usage: pmc_traffic_json.py <dir-with-rd_/wr_counter_collection.csv> <out.json> "<profiled command>"
rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB per dispatch.  On gfx950 FETCH_SIZE tallies 128-byte read requests
as 64 bytes, so reads are doubled: hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (mean per launch of each conv kernel)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if r["Counter_Name"] == counter and ("vits::conv_" in kn or "vits::resblock_pair" in kn or "vits::resblock_chain" in kn):
            name = kn.replace("void ", "").replace("vits::", "").split("(")[0]
            name = re.sub(r"\s+", "", name)
            # the library's timing rows call the staged-input instantiations "...,staged>" and leave "false" out
            # (and the small-grid build "...,staged,6>" / "...,false,6>"; the profiler prints the template arguments
            # "true|false, ring")
            name = re.sub(r",true,(\d+)>", lambda m: ",staged>" if m.group(1) == "0" else f",staged,{m.group(1)}>", name)
            # the fused pair's third template argument is its operand-plane count: 2 = f16x2, 1 = plain bf16
            # (and its fourth the row tiles per wave, which the timing rows leave out)
            name = re.sub(r"resblock_pair_kernel<(\d+),(\d+),2(,\d+)?>", r"resblock_pair_kernel<\1,\2>", name)
            name = re.sub(r"resblock_pair_kernel<(\d+),(\d+),1(,\d+)?>", r"resblock_pair_bf16_kernel<\1,\2>", name)
            # the whole-ResBlock kernel: <channels / 32, taps, planes, ...build parameters>; rows name (channels / 32, taps)
            name = re.sub(r"resblock_chain_kernel<(\d+),(\d+),2(,\d+)*>", r"resblock_chain_kernel<\1,\2>", name)
            name = re.sub(r"resblock_chain_kernel<(\d+),(\d+),1(,\d+)*>", r"resblock_chain_bf16_kernel<\1,\2>", name)
            name = name.replace("conv_glb_kernel<", "conv_glb_bf16_kernel<")     # the timing row names its arithmetic
            name = name.replace(",false,0>", ">").replace(",true>", ",staged>").replace(",false>", ">")
            acc[name].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def main():
    d, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    rd = per_kernel(glob.glob(os.path.join(d, "rd*counter_collection.csv"))[0], "FETCH_SIZE")
    wr = per_kernel(glob.glob(os.path.join(d, "wr*counter_collection.csv"))[0], "WRITE_SIZE")
    kernels = {}
    for k in sorted(rd):
        f = sum(rd[k]) / len(rd[k])
        w = sum(wr[k]) / len(wr[k]) if k in wr else 0.0
        kernels[k] = dict(launches=len(rd[k]), fetch_size_raw=round(f), write_size=round(w), hbm_bytes=round(2 * f + w))
    # the shape the passes ran on: bench.py attaches these bytes to its roofline only when it runs the same shape
    def opt(flag, default):
        m = re.search(flag + r"[ =](\S+)", cmd)
        return type(default)(m.group(1)) if m else default
    workload = dict(batch=opt("--batch", 16), frames=opt("--frames", 258), phonemes=opt("--phonemes", 50),
                    precision=opt("--precision", "f16x2"))
    json.dump(dict(command=cmd, workload=workload,
                   units="bytes per launch (mean over the run's launches of that kernel); hbm_bytes = 2 * FETCH_SIZE + "
                         "WRITE_SIZE: FETCH_SIZE counts 128-byte read requests as 64 bytes on gfx950 (checked on a conv of "
                         "known traffic: reported 306 MB against 541 MB read, 598 MB after doubling; WRITE_SIZE matched "
                         "the 270.5 MB written exactly)",
                   kernels=kernels), open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
