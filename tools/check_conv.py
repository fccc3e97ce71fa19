#!/usr/bin/env python3
"""Spot-check of the conv operator on generator-like shapes against torch.nn.functional.conv1d on the GPU
(development aid for new kernel variants; usage: check_conv.py [fp32|bf16x3|bf16], with VITS_CONV_KS=0 etc.)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402
from vits_amd import weights  # noqa: E402

PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"      # fp32 | bf16x3 | bf16
P = N.PRECISIONS[PREC]


def rounded(t):
    """operand as the kernel sees it: plain bf16 mode rounds both operands to nearest-even bf16"""
    return t.to(torch.bfloat16).to(torch.float32) if PREC == "bf16" else t


torch.manual_seed(0)
stream = torch.cuda.current_stream().cuda_stream
bad = 0
for (C, k, d, L, B) in [(64, 3, 1, 1000, 2), (64, 7, 3, 2500, 3), (128, 11, 5, 700, 2), (128, 3, 1, 4096, 2),
                        (64, 11, 1, 66048, 2), (192, 5, 1, 3000, 2), (64, 3, 1, 256, 40)]:
    w = torch.randn(C, C, k) / math.sqrt(C * k)
    bias = torch.randn(C)
    x = torch.randn(B, C, L)
    res = torch.randn(B, C, L)
    pk = weights.pack_rows(w, P).cuda()
    xd, rd, bd = x.cuda(), res.cuda(), bias.cuda()
    out = torch.empty(B, C, L, device="cuda")
    N.check(N.lib.vits_op_conv1d(pk.data_ptr(), bd.data_ptr(), C, C, k, d, 0, P, xd.data_ptr(), B, L, 0, 0, 0,
                                 rd.data_ptr(), out.data_ptr(), N.F_IN_LRELU | N.F_RES, 1.0, 0.1, stream))
    ref = F.conv1d(rounded(F.leaky_relu(xd, 0.1)).double(), rounded(w).cuda().double(), bd.double(), padding=(k - 1) * d // 2, dilation=d) + rd.double()
    err = (out.double() - ref).abs().max().item()
    print(f"C{C} k{k} d{d} L{L} B{B}: max abs err {err:.3e}")
    bad += err > 1e-4
sys.exit(1 if bad else 0)
