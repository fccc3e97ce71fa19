#!/usr/bin/env python3
"""A ResBlock pair as two staged-input f16x2 conv launches (the generator's path at 64 channels and up), run over the
whole batch at once or over slices of it, conv1 and conv2 of a slice back to back: does the intermediate of a slice
stay in the memory-side cache for the second conv?  usage  python tools/bench_pair_split.py C K L B [iters]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402
from vits_amd import weights  # noqa: E402


def main():
    C, K, L, B = (int(v) for v in sys.argv[1:5])
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 100
    p = N.PRECISIONS["f16x2"]
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    pk1 = weights.pack_rows(torch.randn(C, C, K, generator=g) / math.sqrt(C * K), p).cuda()
    pk2 = weights.pack_rows(torch.randn(C, C, K, generator=g) / math.sqrt(C * K), p).cuda()
    bias = torch.randn(C, generator=g).cuda()
    x = torch.randn(B, C, L, device="cuda")
    xs = torch.empty(B, C, L, device="cuda")                        # staged lrelu(x)
    N.check(N.lib.vits_op_conv1d(pk1.data_ptr(), bias.data_ptr(), C, C, K, 1, 0, p, x.data_ptr(), B, L, 0, 0, 0, 0,
                                 xs.data_ptr(), N.F_IN_LRELU | N.F_OUT_STAGED, 1.0, 0.1, stream))
    mid = torch.empty(B, C, L, device="cuda")
    out = torch.empty(B, C, L, device="cuda")
    row = C * L * 4

    def pair(b0, nb):
        o = b0 * row
        N.check(N.lib.vits_op_conv1d(pk1.data_ptr(), bias.data_ptr(), C, C, K, 1, 0, p, xs.data_ptr() + o, nb, L, 0, 0, 0, 0,
                                     mid.data_ptr() + o, N.F_IN_STAGED | N.F_OUT_STAGED, 1.0, 0.1, stream))
        N.check(N.lib.vits_op_conv1d(pk2.data_ptr(), bias.data_ptr(), C, C, K, 1, 0, p, mid.data_ptr() + o, nb, L, 0, 0, 0,
                                     x.data_ptr() + o, out.data_ptr() + o, N.F_IN_STAGED | N.F_RES, 1.0, 0.1, stream))

    for nsplit in (1, 2, 4, 8, 16):
        if B % nsplit:
            continue
        nb = B // nsplit

        def run():
            for s in range(nsplit):
                pair(s * nb, nb)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"C{C} k{K} L={L} B={B} slices={nsplit:2d} ({nb} rows, {nb * row / 1e6:.0f} MB intermediate each): {us:8.1f} us per pair", flush=True)


if __name__ == "__main__":
    main()
