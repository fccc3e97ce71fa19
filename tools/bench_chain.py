#!/usr/bin/env python3
"""Micro-benchmark of a whole ResBlock (dilations 1, 3, 5) through the C ABI, HIP-event timed: one chain launch
against the same three pairs launched one after the other.
usage  python tools/bench_chain.py PREC C K L B [iters]        (PREC: f16x2 | bf16)"""
import ctypes as Ct
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402
from vits_amd import weights  # noqa: E402


def main():
    prec = sys.argv[1]
    C, K, L, B = (int(v) for v in sys.argv[2:6])
    iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    dils = tuple(int(v) for v in sys.argv[7].split(",")) if len(sys.argv) > 7 else (1, 3, 5)
    n = len(dils)
    p = N.PRECISIONS[prec]
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    pk = [[weights.pack_rows(torch.randn(C, C, K, generator=g) / math.sqrt(C * K), p).cuda() for _ in range(2)]
          for _ in dils]
    bs = [[torch.randn(C, generator=g).cuda() for _ in range(2)] for _ in dils]
    x = torch.randn(B, C, L, device="cuda")
    bufs = [torch.empty(B, C, L, device="cuda") for _ in range(n)]
    arr = lambda ts: (Ct.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
    w1, b1, w2, b2 = arr([a for a, _ in pk]), arr([a for a, _ in bs]), arr([b for _, b in pk]), arr([b for _, b in bs])
    dl = (Ct.c_int * n)(*dils)

    def chain(acc):
        N.check(N.lib.vits_op_resblock_chain(w1, b1, w2, b2, C, K, dl, n, p, x.data_ptr(), bufs[n - 1].data_ptr(), B, L, acc,
                                             0.1, stream))

    def pairs(acc):
        cur = x
        for q in range(n):
            N.check(N.lib.vits_op_resblock_pair_prec(pk[q][0].data_ptr(), bs[q][0].data_ptr(), pk[q][1].data_ptr(),
                                                     bs[q][1].data_ptr(), C, K, dils[q], p, cur.data_ptr(),
                                                     bufs[q].data_ptr(), B, L, acc if q == n - 1 else 0, 0.1, stream))
            cur = bufs[q]

    fl = n * 2.0 * 2 * C * C * K * L * B
    for name, fn in (("chain", chain), ("pairs", pairs)):
        for acc in (0, 1):
            for _ in range(2):
                fn(acc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn(acc)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            by = 4.0 * C * L * B * (2 + acc)                # x once, the result once, the sum when accumulated
            print(f"{name:5s} {prec} C{C} k{K} d{dils} L={L} B={B} acc={acc} {ms * 1e3:9.1f} us {fl / ms / 1e9:7.1f} TF/s "
                  f"{by / ms / 1e6:6.0f} GB/s of ResBlock-minimal bytes", flush=True)


if __name__ == "__main__":
    main()
