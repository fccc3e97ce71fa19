#!/usr/bin/env python3
"""Micro-benchmark of a staged-input f16x2 conv (the second conv of a ResBlock pair) through the C ABI, HIP-event
timed: usage  python tools/bench_gl.py C K DIL L B [flags: res|staged_out|plain] [iters]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402
from vits_amd import weights  # noqa: E402


def main():
    C, K, dil, L, B = (int(v) for v in sys.argv[1:6])
    kind = sys.argv[6] if len(sys.argv) > 6 else "res"
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
    p = N.PRECISIONS["f16x2"]
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    w = torch.randn(C, C, K, generator=g) / math.sqrt(C * K)
    pk = weights.pack_rows(w, p).cuda()
    bias = torch.randn(C, generator=g).cuda()
    x = torch.randn(B, C, L, device="cuda")
    staged = torch.empty(B, C, L, device="cuda")
    N.check(N.lib.vits_op_conv1d(pk.data_ptr(), bias.data_ptr(), C, C, K, dil, 0, p, x.data_ptr(), B, L, 0, 0, 0, 0,
                                 staged.data_ptr(), N.F_IN_LRELU | N.F_OUT_STAGED, 1.0, 0.1, stream))
    out = torch.empty(B, C, L, device="cuda")
    flags = N.F_IN_STAGED | {"res": N.F_RES, "staged_out": N.F_OUT_STAGED, "plain": 0}[kind]
    flags |= int(os.environ.get("VITS_DBG_FLAGS", "0"), 0)
    res = x if kind == "res" else None

    def run():
        N.check(N.lib.vits_op_conv1d(pk.data_ptr(), bias.data_ptr(), C, C, K, 1, 0, p, staged.data_ptr(), B, L, 0, 0, 0,
                                     N.ptr(res), out.data_ptr(), flags, 1.0, 0.1, stream))
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
    fl = 2.0 * C * C * K * L * B
    print(f"C{C} k{K} L={L} B={B} {kind:10s} GL={os.environ.get('VITS_CONV_GL', '1')} WIDE={os.environ.get('VITS_CONV_GL_WIDE', '1')}"
          f" DBG={os.environ.get('VITS_DBG_FLAGS', '0')} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
