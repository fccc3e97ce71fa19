#!/usr/bin/env python3
"""Throughput mode over a few C2-size batches and nothing else: the program to put behind rocprofv3 --kernel-trace
when looking at how the front end and the generator share the GPU (tools/overlap_analyze.py reads the trace).
usage  python tools/overlap_trace.py [batches] [B] [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import SynthesizerTrn, synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 258
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size, **cfg.kwargs())
    net.to("cuda:0").load_state_dict(synthetic.synthetic_state_dict(cfg, include_enc_q=False))
    kws = []
    for i in range(n):
        b = synthetic.make_batch(100 + i, [max(50, frames * 50 // 258)] * B, [frames] * B)
        kws.append({k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in dict(
            phonemes=b.phonemes, phonemes_lengths=b.lengths, sid=b.sid, noise_scale=b.noise_scale,
            duration_control=b.durations, noise=b.noise).items()})
    for _ in net.infer_batches(kws[:3]):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in net.infer_batches(kws):
        pass
    torch.cuda.synchronize()
    print(f"{n} batches of {B} x {frames} frames: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per batch", flush=True)


if __name__ == "__main__":
    main()
