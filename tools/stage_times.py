#!/usr/bin/env python3
"""Per-stage GPU time of one infer call (HIP events around each C-ABI stage call on the current stream).

usage: python tools/stage_times.py [--batch 16] [--frames 258] [--phonemes 50] [--iters 10]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import _native as N  # noqa: E402
from vits_amd import SynthesizerTrn, synthetic  # noqa: E402

STAGES = ("vits_speaker_cond", "vits_text_encoder", "vits_variance_adaptor", "vits_length_regulate_plan",
          "vits_length_regulate", "vits_frame_prior", "vits_flow_reverse", "vits_generator")


class TimedLib:
    def __init__(self, lib):
        self._lib = lib
        self.events = collections.defaultdict(list)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in STAGES:
            return fn

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            self.events[name].append((e0, e1))
            return rc
        return timed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=258)
    ap.add_argument("--phonemes", type=int, default=50)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         **cfg.kwargs()).eval().to(dev)
    net.load_state_dict(synthetic.synthetic_state_dict(cfg, include_enc_q=False), strict=True)
    b = synthetic.make_batch(1, [a.phonemes] * a.batch, [a.frames] * a.batch)
    d = dict(phonemes=b.phonemes.to(dev), lengths=b.lengths.to(dev), sid=b.sid.to(dev), durations=b.durations.to(dev),
             noise=b.noise.to(dev))

    def step():
        return net.infer(d["phonemes"], d["lengths"], sid=d["sid"], noise_scale=0.667, duration_control=d["durations"],
                         noise=d["noise"])[0]
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    timed = TimedLib(N.lib)
    N.lib = timed
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.iters):
        step()
    t1.record()
    torch.cuda.synchronize()
    total = t0.elapsed_time(t1) / a.iters
    acc = 0.0
    for name in STAGES:
        ms = sum(e0.elapsed_time(e1) for e0, e1 in timed.events[name]) / a.iters
        acc += ms
        print(f"{name:28s} {ms:8.3f} ms  {100 * ms / total:5.1f} %")
    print(f"{'sum of stages':28s} {acc:8.3f} ms; whole step {total:.3f} ms")


if __name__ == "__main__":
    main()
