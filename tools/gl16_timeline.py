#!/usr/bin/env python3
"""Where the time of a conv_gl16_kernel launch goes, block by block: 100 MHz stamps at block entry, main-loop start,
main-loop end and block end (stores accepted), plus the compute unit each block ran on.  Needs the diagnostic build:

  tools/build_variant.sh timeline kernels_gl16.hip -DGL16_TIMELINE
  VITS_AMD_LIB=$PWD/vits_amd/lib/variants/libvits_timeline.so python3 tools/gl16_timeline.py C K L B [res|staged_out]

Prints one JSON line for the last of 30 back-to-back launches: the launch's span, the median prologue / loop / epilogue
of a block, how many blocks shared a compute unit at a time, and the matrix-pipe view: the share of the span in which
0, 1 and 2 blocks of a compute unit were inside their main loops (averaged over the compute units)."""
import ctypes as C
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402
from vits_amd import weights  # noqa: E402


def main():
    Cc, K, L, B = (int(v) for v in sys.argv[1:5])
    kind = sys.argv[5] if len(sys.argv) > 5 else "res"
    fn = getattr(N.lib, "vits_debug_gl16_timeline_buffer", None)
    if fn is None:
        raise SystemExit("this library has no timeline stamps: tools/build_variant.sh timeline kernels_gl16.hip -DGL16_TIMELINE")
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
    buf = torch.zeros(slots, 6, dtype=torch.int64, device="cuda")
    assert fn(buf.data_ptr(), slots) == 0
    flags = N.F_IN_STAGED | (N.F_RES if kind == "res" else N.F_OUT_STAGED)
    res = x if kind == "res" else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        N.check(N.lib.vits_op_conv1d(pk.data_ptr(), bias.data_ptr(), Cc, Cc, K, 1, 0, p, staged.data_ptr(), B, L, 0, 0, 0,
                                     N.ptr(res), out.data_ptr(), flags, 1.0, 0.1, stream))
    e1.record()
    torch.cuda.synchronize()
    st = buf.cpu().numpy()
    st = st[st[:, 3] > 0]
    t = st[:, :4].astype(np.float64)
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0                                            # microseconds from the first block's entry
    span = t[:, 3].max()
    hw, xcc = st[:, 4], st[:, 5] & 0xF
    cu = ((xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)).astype(np.int64)
    cus = np.unique(cu)
    # per compute unit: time with 0 / 1 / 2+ blocks inside their main loops
    share = np.zeros(4)
    resident_max = 0
    for c in cus:
        sel = t[cu == c]
        ev = sorted([(v, 1) for v in sel[:, 1]] + [(v, -1) for v in sel[:, 2]])
        level, last = 0, 0.0
        for when, d in ev:
            share[min(level, 3)] += when - last
            level, last = level + d, when
        share[0] += span - last
        ev = sorted([(v, 1) for v in sel[:, 0]] + [(v, -1) for v in sel[:, 3]])
        level = 0
        for _, d in ev:
            level += d
            resident_max = max(resident_max, level)
    share /= span * len(cus)
    med = lambda v: round(float(np.median(v)), 2)  # noqa: E731
    order = np.argsort(t[:, 0])
    print(json.dumps({
        "shape": {"C": Cc, "K": K, "L": L, "B": B, "kind": kind}, "blocks": int(len(t)), "compute_units_seen": int(len(cus)),
        "event_avg_launch_us": round(e0.elapsed_time(e1) / 30 * 1e3, 1), "span_us": round(float(span), 1),
        "prologue_us_median": med(t[:, 1] - t[:, 0]), "loop_us_median": med(t[:, 2] - t[:, 1]),
        "loop_us_p10_p90": [round(float(np.percentile(t[:, 2] - t[:, 1], q)), 1) for q in (10, 90)],
        "epilogue_us_median": med(t[:, 3] - t[:, 2]),
        "epilogue_us_p10_p90": [round(float(np.percentile(t[:, 3] - t[:, 2], q)), 1) for q in (10, 90)],
        "max_blocks_resident_per_cu": int(resident_max),
        "share_of_span_with_n_blocks_in_loop": {str(i): round(float(share[i]), 3) for i in range(4)},
        "entry_us_deciles": [round(float(v), 1) for v in np.percentile(t[:, 0], range(0, 101, 10))],
        "first_blocks_entry_us": [round(float(v), 2) for v in t[order[:6], 0]],
    }))


if __name__ == "__main__":
    main()
