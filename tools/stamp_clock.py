#!/usr/bin/env python3
"""The shader clock the chip holds inside the main loop of the LDS-DMA conv (conv_gl16_kernel), from in-kernel stamps:
shader-cycle counter / 100 MHz wall counter around the loop of every block, after at least two seconds of
back-to-back launches on random data.  Needs the diagnostic build of the library:

  tools/build_variant.sh stamp kernels_gl16.hip -DGL16_STAMP
  VITS_AMD_LIB=$PWD/vits_amd/lib/variants/libvits_stamp.so python3 tools/stamp_clock.py [C K L B [seconds]]

Prints one JSON line: median / 10th / 90th percentile clock over the blocks of the last launch, and the event-timed
average launch duration of the same run."""
import ctypes as C
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402
from vits_amd import weights  # noqa: E402


def main():
    Cc, K, L, B = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (128, 11, 65536, 16)
    seconds = float(sys.argv[5]) if len(sys.argv) > 5 else 2.5
    fn = getattr(N.lib, "vits_debug_gl16_stamp_buffer", None)
    if fn is None:
        raise SystemExit("this library has no stamps: build it with tools/build_variant.sh stamp kernels_gl16.hip -DGL16_STAMP")
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_uint64]
    p = N.PRECISIONS["f16x2"]
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    pk = weights.pack_rows(torch.randn(Cc, Cc, K, generator=g) / math.sqrt(Cc * K), p).cuda()
    bias = torch.randn(Cc, generator=g).cuda()
    x = torch.randn(B, Cc, L, device="cuda")
    staged = torch.empty(B, Cc, L, device="cuda")
    N.check(N.lib.vits_op_conv1d(pk.data_ptr(), bias.data_ptr(), Cc, Cc, K, 1, 0, p, x.data_ptr(), B, L, 0, 0, 0, 0,
                                 staged.data_ptr(), N.F_IN_LRELU | N.F_OUT_STAGED, 1.0, 0.1, stream))
    out = torch.empty(B, Cc, L, device="cuda")
    slots = 1 << 16
    stamps = torch.zeros(slots, 2, dtype=torch.int64, device="cuda")
    assert fn(stamps.data_ptr(), slots) == 0

    def run():
        N.check(N.lib.vits_op_conv1d(pk.data_ptr(), bias.data_ptr(), Cc, Cc, K, 1, 0, p, staged.data_ptr(), B, L, 0, 0, 0,
                                     x.data_ptr(), out.data_ptr(), N.F_IN_STAGED | N.F_RES, 1.0, 0.1, stream))
    N.lib.vits_profile_reset()
    N.lib.vits_profile_enable(1)
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
        n += 50
    N.lib.vits_profile_enable(0)
    rows = N.profile_rows()
    st = stamps.cpu()
    st = st[st[:, 1] > 0].double()
    clk = (st[:, 0] / st[:, 1] * 0.1).sort().values             # GHz: cycles per 10 ns tick
    q = lambda f: float(clk[min(len(clk) - 1, int(f * len(clk)))])  # noqa: E731
    print(json.dumps({"shape": {"C": Cc, "K": K, "L": L, "B": B}, "launches": n, "seconds": round(time.time() - t0, 2),
                      "blocks_stamped": len(clk), "clock_ghz_median": round(q(0.5), 3), "clock_ghz_p10": round(q(0.1), 3),
                      "clock_ghz_p90": round(q(0.9), 3),
                      "loop_us_median": round(float(st[:, 1].sort().values[len(st) // 2]) / 100.0, 1),
                      "kernels": [{"kernel": r["name"], "avg_us": round(r["ms"] * 1e3 / r["count"], 1),
                                   "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1)} for r in rows]}))


if __name__ == "__main__":
    main()
