#!/usr/bin/env python3
"""Micro-benchmark of the conv kernel on the generator's layer shapes (through the C ABI, HIP-event timed).

usage: python tools/bench_conv.py [--batch 16] [--frames 258] [--iters 5] [--only SUBSTR]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402
from vits_amd import weights  # noqa: E402


def layer_list(T):
    out = [("conv_pre 192->512 k7", 512, 192, 7, 1, T, 0, 1)]
    ch, L = 512, T
    for i, (u, ku) in enumerate(zip((8, 8, 4, 2), (16, 16, 4, 4))):
        cin, ch = ch, ch // 2
        mode = {8: N.MODE_UP8, 4: N.MODE_UP4, 2: N.MODE_UP2}[u]
        out.append((f"ups{i} {cin}->{ch} u{u}", ch * u, cin, ku // u, 1, L, mode, 1))
        L *= u
        for k in (3, 7, 11):
            for d in (1, 3, 5):
                out.append((f"rb C{ch} k{k} d{d}", ch, ch, k, d, L, 0, 1 if d > 1 else 4))   # d=1 occurs 4x (c1 d1 + 3 c2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=258)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--prec", default="bf16x3")
    a = ap.parse_args()
    B = a.batch
    stream = torch.cuda.current_stream().cuda_stream
    tot_t = tot_f = 0.0
    for name, rows, cin, taps, dil, L, mode, mult in layer_list(a.frames):
        if a.only and a.only not in name:
            continue
        up = {N.MODE_UP8: 8, N.MODE_UP4: 4, N.MODE_UP2: 2}.get(mode, 1)
        w = torch.randn(rows, cin, taps) / math.sqrt(cin * taps)
        pk = weights.pack_rows(w, N.PRECISIONS[a.prec]).cuda()
        bias = torch.randn(rows).cuda()
        x = torch.randn(B, cin, L, device="cuda")
        res = torch.randn(B, rows // up, L, device="cuda") if mode == 0 else None
        out = torch.empty(B, rows // up, L * up, device="cuda")
        flags = (N.F_IN_LRELU | N.F_RES) if mode == 0 else N.F_IN_LRELU
        flags |= int(os.environ.get("VITS_DBG_FLAGS", "0"))

        def run():
            N.check(N.lib.vits_op_conv1d(pk.data_ptr(), bias.data_ptr(), cin, rows, taps, dil, mode, N.PRECISIONS[a.prec],
                                         x.data_ptr(), B, L,
                                         0, 0, 0, N.ptr(res), out.data_ptr(), flags, 1.0, 0.1, stream))
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 2.0 * rows * cin * taps * L * B
        tot_t += ms * mult
        tot_f += fl * mult
        print(f"{name:24s} L={L:7d} {ms * 1e3:9.1f} us {fl / ms / 1e9:7.1f} TF/s  x{mult}")
    print(f"weighted total: {tot_t:.3f} ms, {tot_f / tot_t / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
