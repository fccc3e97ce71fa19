#!/usr/bin/env python3
"""Generator-only timing on a fixed latent (C2 shape by default): the front end runs once, then vits_generator is
called back to back, and with a host wait after every call, with 1 / 2 / 3 chain streams.  usage  python tools/bench_generator.py [B] [frames] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vits_amd import SynthesizerTrn, synthetic  # noqa: E402
from vits_amd import _native as N  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 258
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size, **cfg.kwargs())
    net.to("cuda:0").load_state_dict(synthetic.synthetic_state_dict(cfg, include_enc_q=False))
    b = synthetic.make_batch(5, [50] * B, [frames] * B)
    with torch.no_grad():
        fe = net._frontend(b.phonemes, b.lengths, b.sid, b.noise_scale, None, None, None, b.durations, b.noise, None)
    lib, h = N.lib, net._handle
    T = fe["T"]
    ws = torch.empty(int(lib.vits_generator_workspace_bytes(h, B, T)), dtype=torch.uint8, device="cuda")
    o = torch.empty(B, 1, T * 512, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def gen():
        N.check(lib.vits_generator(h, fe["z"].data_ptr(), fe["totals"].data_ptr(), fe["cond"].data_ptr(), B, T, T,
                                   o.data_ptr(), ws.data_ptr(), ws.numel(), s), "generator")
    for streams in (1, 2, 3):
        N.check(lib.vits_model_set_generator_streams(h, streams), "streams")
        for _ in range(3):
            gen()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            gen()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        # the same with the host waiting for every call before it enqueues the next (what a pipeline that allows one
        # generator call in flight pays between two calls)
        t0 = time.perf_counter()
        for _ in range(iters):
            gen()
            torch.cuda.synchronize()
        ms_sync = (time.perf_counter() - t0) / iters * 1e3
        print(f"generator B={B} T={T} chain streams={streams}: {ms:.3f} ms per call back to back, {ms_sync:.3f} ms with a "
              f"host wait after each", flush=True)


if __name__ == "__main__":
    main()
