#!/usr/bin/env python3
"""Micro-benchmark of the attention operator through the C ABI, HIP-event timed: usage
python tools/bench_attention.py B L [iters] [bf16x3|f16x2]   (VITS_ATT_MFMA=<min length> moves the matrix-core kernel's
threshold; the last argument chooses that kernel's operand split)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402


def main():
    B, L = int(sys.argv[1]), int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    prec = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
    H, dk = 2, 96
    C = H * dk
    qkv = torch.randn(B, 3 * C, L, device="cuda")
    ek, ev = torch.randn(9, dk, device="cuda") * 0.1, torch.randn(9, dk, device="cuda") * 0.1
    lengths = torch.full((B,), L, dtype=torch.int32, device="cuda")
    out = torch.empty(B, C, L, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        N.check(N.lib.vits_op_attention_prec(qkv.data_ptr(), ek.data_ptr(), ev.data_ptr(), lengths.data_ptr(), out.data_ptr(),
                                             B, H, dk, L, 4, N.PRECISIONS[prec], stream))
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
    fl = 4.0 * B * C * L * L                     # q.k and p.v, dense
    print(f"attention {prec} B={B} L={L} VITS_ATT_MFMA={os.environ.get('VITS_ATT_MFMA', 'default')} {us:9.1f} us "
          f"{fl / us / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
