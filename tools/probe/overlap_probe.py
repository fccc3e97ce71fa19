#!/usr/bin/env python3
"""Do a compute-bound and a memory-bound conv launch overlap when issued on two streams?"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import _native as N, weights

def mk(rows, cin, taps, L, B=16):
    w = torch.randn(rows, cin, taps) / math.sqrt(cin * taps)
    pk = weights.pack_rows(w, 1).cuda()
    return dict(pk=pk, bias=torch.randn(rows).cuda(), x=torch.randn(B, cin, L, device="cuda"),
                res=torch.randn(B, rows, L, device="cuda"), out=torch.empty(B, rows, L, device="cuda"),
                rows=rows, cin=cin, taps=taps, L=L, B=B)

def run(c, stream):
    N.check(N.lib.vits_op_conv1d(c["pk"].data_ptr(), c["bias"].data_ptr(), c["cin"], c["rows"], c["taps"], 1, 0, 1,
                                 c["x"].data_ptr(), c["B"], c["L"], 0, 0, 0, c["res"].data_ptr(), c["out"].data_ptr(),
                                 N.F_IN_LRELU | N.F_RES, 1.0, 0.1, stream.cuda_stream))

a, b = mk(128, 128, 11, 16512), mk(32, 32, 3, 132096)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
ta = timeit(lambda: [run(a, s1) for _ in range(4)])
tb = timeit(lambda: [run(b, s1) for _ in range(12)])
tc = timeit(lambda: ([run(a, s1) for _ in range(4)], [run(b, s2) for _ in range(12)]))
ti = timeit(lambda: [(run(a, s1), run(b, s2), run(b, s2), run(b, s2)) for _ in range(4)])
print(f"4x compute-bound alone {ta:.3f} ms; 12x memory-bound alone {tb:.3f} ms; sum {ta+tb:.3f}; two streams {tc:.3f} ms; interleaved issue {ti:.3f} ms")
