#!/usr/bin/env python3
"""Micro-benchmark of the fused ResBlock pair through the C ABI, HIP-event timed:
usage  python tools/bench_pair.py C K DIL L B [iters]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402
from vits_amd import weights  # noqa: E402


def main():
    C, K, dil, L, B = (int(v) for v in sys.argv[1:6])
    iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    p = N.PRECISIONS["f16x2"]
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    w1 = weights.pack_rows(torch.randn(C, C, K, generator=g) / math.sqrt(C * K), p).cuda()
    w2 = weights.pack_rows(torch.randn(C, C, K, generator=g) / math.sqrt(C * K), p).cuda()
    b1, b2 = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    x = torch.randn(B, C, L, device="cuda")
    out = torch.empty(B, C, L, device="cuda")

    def run():
        N.check(N.lib.vits_op_resblock_pair(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), C, K, dil,
                                            x.data_ptr(), out.data_ptr(), B, L, 0, 0.1, stream))
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * 2 * C * C * K * L * B
    print(f"pair C{C} k{K} d{dil} L={L} B={B} lib={os.path.basename(os.environ.get('VITS_AMD_LIB', 'default'))}"
          f" {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s  {2 * 4.0 * C * L * B / ms / 1e6:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
