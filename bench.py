#!/usr/bin/env python3
"""Benchmark of the MVTS synthesis path (SynthesizerTrn.infer) on MI355X.

  python bench.py --gpus N --steps K --warmup W

Started plainly with N > 1 it launches its own N ranks (one fresh process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, before anything in the parent touches a GPU), the way the reference's multi-GPU entry spawns its
workers (reference train.py:44-48); started under torch.distributed.run it is one of the ranks.

A step is one infer call over one batch of synthetic utterances whose inputs are already resident in HBM.  The
workload is BASELINE.json's configs[1]: 16 utterances of 50 phonemes / 258 frames (3.0 s of 44.1 kHz audio each),
fixed MFA-style durations, fixed noise; every rank runs its own batch (utterance-level data parallelism, no
collective on the data path, weak scaling).  Rank 0 prints one JSON line:
  value      = audio-seconds synthesised per wall-second, summed over ranks (max-over-ranks time)
  roofline   = the dominant kernel (the MFMA conv instantiation with the largest summed time) timed with HIP events on
               its launch stream; algorithmic fp32 FLOP/s against the dense MFMA peak of the mode it runs in
               (fp32 MFMA: 157.3 TFLOP/s; f16x2: three fp16 products per fp32 product, 2500 / 3 = 833.3 TFLOP/s;
               bf16x3: six bf16 products, 2500 / 6 = 416.7 TFLOP/s; bf16: 2500), or its algorithmic bytes against
               8 TB/s when its arithmetic intensity puts it under the HBM roof
  cpu_baseline = the CPU oracle (a PyTorch fp32 port of the reference's infer) timed on this host's cores on a
                 bounded sample of the same workload
  sub_results  = (one GPU, default workload only) the other BASELINE configurations, each with its own ms_per_step and
                 roofline (10 steps after 2 warm-up steps): C3 (64 x 10 s, bf16 generator convs), the C3 shape in the
                 default f16x2 arithmetic, a fixed sample of the C4 corpus, and C2 in bf16x3 arithmetic
  config       = the workload, and with it the batch-1 figures of BASELINE's metric: rtf_batch1, batch1_latency_ms (C1),
                 batch1_edit_latency_ms (C5), stream_first_window_ms
The per-kernel table of every roofline pass goes to stderr as JSON lines of its own; the result line stays a few KB.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X dense fp32 matrix peak (256 CUs x 256 FLOP/clk x 2.4 GHz)
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X dense bf16 / fp16 matrix peak
PEAK_HBM_GBPS = 8000.0             # MI355X HBM3E
# bf16x3 kernels spend six bf16 MFMA products per algorithmic fp32 product
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
# f16x2 kernels spend three fp16 MFMA products (same dense rate as bf16) per algorithmic fp32 product
PEAK_F16X2_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0


def build_model(device, precision=None):
    from vits_amd import SynthesizerTrn, synthetic
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         precision=precision, **cfg.kwargs()).eval()
    net.to(device)
    net.load_state_dict(synthetic.synthetic_state_dict(cfg, include_enc_q=False), strict=True)
    return net


def to_device(batch, device):
    return dict(phonemes=batch.phonemes.to(device), lengths=batch.lengths.to(device), sid=batch.sid.to(device),
                durations=batch.durations.to(device), noise=batch.noise.to(device))


def batch_kwargs(d, noise_scale=0.667):
    return dict(phonemes=d["phonemes"], phonemes_lengths=d["lengths"], sid=d["sid"], noise_scale=noise_scale,
                duration_control=d["durations"], noise=d["noise"])


def run_steps(net, batches, overlap):
    """One infer per batch.  overlap: SynthesizerTrn.infer_batches (the front end of batch i+1 runs beside the
    generator of batch i on a second stream); otherwise plain back-to-back infer calls."""
    if overlap:
        for _ in net.infer_batches(batch_kwargs(d) for d in batches):
            pass
    else:
        for d in batches:
            step(net, d)


def step(net, d, noise_scale=0.667):
    return net.infer(d["phonemes"], d["lengths"], sid=d["sid"], noise_scale=noise_scale,
                     duration_control=d["durations"], noise=d["noise"])[0]


def algorithmic_flops(phoneme_counts, frame_counts, predicted_duration=False):
    """2*MAC per utterance, banded relative attention, weight-norm folded (duration predictor only when it runs)."""
    per_p = 25757696 - (0 if predicted_duration else 688640)
    return sum(837861376 * t + 3072 * t * t + per_p * p + 7680 * p * p for p, t in zip(phoneme_counts, frame_counts))


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` (2 * FETCH_SIZE + WRITE_SIZE) from the committed rocprofv3 --pmc passes over
    this benchmark's command (profiles/*_pmc_traffic.json, newest first: the command, the gfx950 FETCH_SIZE correction
    and its check are recorded there); None when no pass covers the kernel or the passes ran on another shape or
    precision."""
    import glob
    for name in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            with open(name) as f:
                table = json.load(f)
            if table["workload"] != workload:
                continue
            return table["kernels"][kernel]["hbm_bytes"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def kernel_peak_tflops(name):
    if name.startswith("conv_mfma_kernel"):
        return PEAK_FP32_MFMA_TFLOPS
    if name.startswith(("conv_bf16_kernel", "conv_glb_bf16_kernel", "resblock_pair_bf16_kernel", "resblock_chain_bf16_kernel")):
        return PEAK_BF16_MFMA_TFLOPS
    if name.startswith(("conv_f16x2_kernel", "conv_gl_", "conv_gl16_", "resblock_pair_kernel", "resblock_chain_kernel")):
        return PEAK_F16X2_TFLOPS
    return PEAK_BF16X3_TFLOPS


def roofline_pass(net, run_once, steps, device, traffic_key=None, top=3, label=None):
    """Per-launch HIP-event timing of every conv kernel over `steps` calls of run_once() (plain infer calls: the
    library keeps the generator on one stream while timing is on, so intervals do not overlap); returns the roofline
    object of the instantiation with the largest summed time.  The object names the `top` kernels by summed time; the
    whole per-kernel table goes to stderr as a JSON line of its own, so that the result line stays a few kilobytes."""
    from vits_amd import _native as N
    N.lib.vits_profile_reset()
    N.lib.vits_profile_enable(1)
    for _ in range(steps):
        run_once()
    torch.cuda.synchronize(device)
    N.lib.vits_profile_enable(0)
    rows = sorted(N.profile_rows(), key=lambda r: -r["ms"])
    N.lib.vits_profile_reset()
    dom = rows[0]                       # rows are named like the kernel symbols a profiler's table lists
    conv_ms = sum(r["ms"] for r in rows)
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    peak = kernel_peak_tflops(dom["name"])
    gbps = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
    # the bound is whichever roof the kernel's arithmetic intensity puts lower: the plain-bf16 generator layers
    # (64 / 32 channels) sit under the HBM roof, the fp32-grade modes under the MFMA roof
    if dom["flops"] / dom["bytes"] < peak * 1e12 / (PEAK_HBM_GBPS * 1e9):
        roof = dict(bound="hbm", kernel=dom["name"], achieved=round(gbps, 1), peak=PEAK_HBM_GBPS, unit="GB/s",
                    frac=round(gbps / PEAK_HBM_GBPS, 4), tflops=round(achieved, 3), mfma_peak_tflops=round(peak, 1),
                    mfma_frac=round(achieved / peak, 4))
    else:
        roof = dict(bound="mfma", kernel=dom["name"], achieved=round(achieved, 3), peak=round(peak, 1),
                    unit="TFLOP/s", frac=round(achieved / peak, 4), algorithmic_gbps=round(gbps, 1))
    roof.update(traffic=pmc_traffic(dom["name"], traffic_key) if traffic_key else None,
                algorithmic_bytes_per_launch=round(dom["bytes"] / dom["count"]),
                launches_per_step=round(dom["count"] / steps, 2),
                avg_launch_us=round(dom["ms"] * 1e3 / dom["count"], 2),
                flops_per_launch=round(dom["flops"] / dom["count"]),
                share_of_conv_time=round(dom["ms"] / conv_ms, 3))
    kernels = [dict(kernel=r["name"], tflops=round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                    gbps=round(r["bytes"] / (r["ms"] * 1e-3) / 1e9), ms_per_step=round(r["ms"] / steps, 3),
                    launches_per_step=round(r["count"] / steps, 2)) for r in rows]
    if top:
        roof["top_kernels"] = kernels[:top]
    print(json.dumps({"conv_kernel_table": label or "headline", "steps": steps, "kernels": kernels}), file=sys.stderr,
          flush=True)
    return roof


def cpu_baseline(phonemes=50, frames=258, sample_utts=4, runs=2):
    """Time the CPU oracle on a bounded sample: `sample_utts` utterances of the benchmark workload's shape."""
    from oracle.mvts_oracle import OracleSynthesizer
    from vits_amd import synthetic
    # the GPU box gives a one-GPU job 16 CPUs of a much larger host; os.cpu_count() reports the host's
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    sd = synthetic.synthetic_state_dict(include_enc_q=False)
    orc = OracleSynthesizer(sd)
    b = synthetic.make_batch(1, [phonemes] * sample_utts, [frames] * sample_utts)
    kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
    orc.infer(b.phonemes[:1], b.lengths[:1], sid=b.sid[:1], noise_scale=b.noise_scale,
              duration_control=b.durations[:1], noise=b.noise[:1])           # warm-up
    best = float("inf")
    out = None
    for _ in range(runs):
        t0 = time.perf_counter()
        out = orc.infer(b.phonemes, b.lengths, **kw)[0]
        best = min(best, time.perf_counter() - t0)
    return dict(value=round(b.audio_seconds / best, 3), unit="audio-s/s", cores=torch.get_num_threads(),
                host_cores=os.cpu_count(), kind="port",
                sample=f"{sample_utts} utterances x {frames} frames of the benchmark workload, fp32, best of {runs} "
                       f"runs after warm-up; PyTorch CPU oracle (oracle/mvts_oracle.py)"), out, b


def dtype_of(net):
    """The arithmetic the dominant work runs in (not a precision claim): f16x2 = fp32 operands split into two fp16
    terms, three fp16 MFMA products, fp32 accumulate; bf16x3 = three bf16 terms, six products; fp32 = fp32 MFMA."""
    return {"f16x2": "f16x2 (fp32 operands as two fp16 terms, fp32 accumulate)",
            "bf16x3": "bf16x3 (fp32 operands as three bf16 terms, fp32 accumulate)",
            "fp32": "f32", "bf16": "bf16"}[net.precision]


def corpus_data(device, utterances, max_batch, world, rank, every=1):
    from vits_amd import sharding, synthetic
    batches = synthetic.corpus_c4(utterances, max_batch=max_batch)[::every]
    mine = sharding.deal_batches(batches, world, rank)
    data = [to_device(synthetic.make_batch(1000 + i, *batches[i]), device) for i in mine]
    return batches, mine, data


def run_corpus(net, args, device, rank, world, dist, red_dev):
    """BASELINE configs[3]: a synthetic corpus (lognormal lengths, 88..1400 frames) sorted by length, cut into
    batches of <= 64 utterances whose composition does not depend on the rank count, whole batches dealt to ranks
    longest-first; every rank synthesises its own batches with no collective on the data path.  Inputs of all of a
    rank's batches are resident in HBM before the clock starts.  value = corpus audio-seconds / slowest rank's time."""
    from vits_amd import sharding
    batches, mine, data = corpus_data(device, args.utterances, args.batch if args.batch != 16 else 64, world, rank)
    # warm-up on the rank's largest batch first (sizes the workspace once), then on its smallest
    order = sorted(range(len(mine)), key=lambda j: -sharding.batch_cost(batches[mine[j]]))
    overlap = not args.no_overlap
    run_steps(net, [data[j] for j in ([order[0], order[-1]] if order else [])], overlap)
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run_steps(net, data, overlap)
    torch.cuda.synchronize(device)
    mine_s = time.perf_counter() - t0
    elapsed = mine_s
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)      # reporting only: the slowest rank's wall time
        elapsed = float(t.item())
    if rank == 0:
        frames = sum(sum(ts) for _, ts in batches)
        audio_s = frames * 512 / 44100.0
        flops = sum(algorithmic_flops(ps, ts) for ps, ts in batches)
        result = {
            "metric": "audio-sec/sec at 44.1 kHz", "value": round(audio_s / elapsed, 2), "unit": "audio-s/s",
            "n_gpus": world, "steps": len(mine), "warmup": 2, "ms_per_step": round(mine_s / max(len(mine), 1) * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": dtype_of(net), "conv_precision": net.precision, "data": "synthetic",
            "config": {"workload": f"C4: corpus of {args.utterances} utterances ({audio_s:.0f} audio-s, {frames} frames, "
                                   f"88..1400 frames each), {len(batches)} length-sorted batches of <= 64 dealt "
                                   "longest-first to the ranks; a step is one batch",
                       "utterances": args.utterances, "batches": len(batches), "parallelism": f"utterance-dp{world}"},
            "whole_path_tflops": round(flops / elapsed / 1e12, 2), "seconds": round(elapsed, 3)}
        # dominant kernel over rank 0's own batches (one event-timed pass over a spread of them)
        sample = data[::max(1, len(data) // 8)]
        it = iter(sample * 1000)
        result["roofline"] = roofline_pass(net, lambda: step(net, next(it)), len(sample), device, label="C4")
        if world == 1 and not args.no_cpu_baseline:
            cb, _, _ = cpu_baseline(phonemes=35, frames=386, sample_utts=3)     # the corpus's median utterance
            result["cpu_baseline"] = cb
        print(json.dumps(result), flush=True)


def timed_steps(net, d, steps, warmup, overlap, device):
    run_steps(net, [d] * warmup, overlap)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run_steps(net, [d] * steps, overlap)
    torch.cuda.synchronize(device)
    return time.perf_counter() - t0


def sub_results(net, device, overlap):
    """The other BASELINE configurations on one GPU, short runs: each entry has value, ms_per_step and its own
    roofline (dominant kernel of THAT workload)."""
    from vits_amd import synthetic
    out = []

    def batch_entry(name, model, batch, steps, warmup):
        d = to_device(batch, device)
        el = timed_steps(model, d, steps, warmup, overlap, device)
        B, P, T = batch.phonemes.shape[0], int(batch.lengths.max()), int(batch.frames.max())
        roof = roofline_pass(model, lambda: step(model, d), max(2, steps // 2), device, top=0, label=name,
                             traffic_key=dict(batch=B, frames=T, phonemes=P, precision=model.precision))
        flops = algorithmic_flops(batch.lengths.tolist(), batch.frames.tolist())
        out.append({"workload": name, "value": round(steps * batch.audio_seconds / el, 2), "unit": "audio-s/s",
                    "ms_per_step": round(el / steps * 1e3, 3), "steps": steps, "warmup": warmup,
                    "conv_precision": model.precision, "shape": [B, P, T],
                    "whole_path_tflops": round(steps * flops / el / 1e12, 2), "roofline": roof})

    c3 = synthetic.config_c3()

    def other_precision(prec, name, batch, steps, warmup):
        model = build_model(device, prec)
        batch_entry(name, model, batch, steps, warmup)
        del model
        torch.cuda.empty_cache()

    # BASELINE configs[2] as quoted comes first, then the same shape in the default arithmetic
    other_precision("bf16", "C3 (64 x 10 s), bf16 generator convs", c3, 10, 2)
    batch_entry("C3 shape (64 x 10 s) in f16x2", net, c3, 10, 2)
    # a fixed sample of the C4 corpus: every 5th of its 157 length-sorted batches (shortest and longest included)
    batches, mine, data = corpus_data(device, 10000, 64, 1, 0, every=5)
    run_steps(net, [data[-1], data[0]], overlap)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run_steps(net, data, overlap)
    torch.cuda.synchronize(device)
    el = time.perf_counter() - t0
    frames = sum(sum(ts) for _, ts in batches)
    it = iter(data[::4] * 1000)
    out.append({"workload": f"C4 sample: every 5th of the corpus's 157 length-sorted batches ({len(batches)} batches, "
                            f"{sum(len(ts) for _, ts in batches)} utterances of 88..1400 frames)",
                "value": round(frames * 512 / 44100.0 / el, 2), "unit": "audio-s/s",
                "ms_per_step": round(el / len(data) * 1e3, 3), "steps": len(data), "warmup": 2,
                "conv_precision": net.precision,
                "whole_path_tflops": round(sum(algorithmic_flops(ps, ts) for ps, ts in batches) / el / 1e12, 2),
                "roofline": roofline_pass(net, lambda: step(net, next(it)), len(data[::4]), device, top=0,
                                          label="C4 sample")})
    del data
    other_precision("bf16x3", "C2 in bf16x3 (full fp32 range)", synthetic.config_c2(), 10, 2)
    return out


def batch1_latencies(net, device):
    """Medians of 20 synchronised calls: C1 (configs[0]'s utterance: 50 phonemes, 3.0 s), C5 (configs[4]: the same
    utterance with per-phoneme F0 / energy / duration overrides), and the time to the first 32-frame window of a
    streamed C1 call.  Measured before the throughput run creates its two pipeline streams: a latency-serving process
    does not run infer_batches, and the generator's chains then have three streams instead of two."""
    from vits_amd import synthetic
    b1 = to_device(synthetic.config_c1(), device)
    for _ in range(3):
        step(net, b1)
    torch.cuda.synchronize(device)
    lat = []
    for _ in range(20):
        t1 = time.perf_counter()
        step(net, b1)
        torch.cuda.synchronize(device)
        lat.append(time.perf_counter() - t1)
    lat.sort()
    c5 = synthetic.config_c5()
    d5 = to_device(c5, device)
    kw5 = dict(sid=d5["sid"], noise_scale=0.667, duration_control=d5["durations"], noise=d5["noise"],
               pitch_control=c5.pitch_hz.to(device), energy_control=c5.energy.to(device))
    lat5 = []
    for i in range(23):
        t1 = time.perf_counter()
        net.infer(d5["phonemes"], d5["lengths"], **kw5)
        torch.cuda.synchronize(device)
        if i >= 3:
            lat5.append(time.perf_counter() - t1)
    lat5.sort()
    kw1 = dict(sid=b1["sid"], noise_scale=0.667, duration_control=b1["durations"], noise=b1["noise"])
    lat_first = []
    for i in range(23):
        t1 = time.perf_counter()
        next(iter(net.infer_stream(b1["phonemes"], b1["lengths"], chunk_frames=32, **kw1)))
        torch.cuda.synchronize(device)
        if i >= 3:
            lat_first.append(time.perf_counter() - t1)
    lat_first.sort()
    med = lat[len(lat) // 2]
    return {"rtf_batch1": round(med / synthetic.config_c1().audio_seconds, 5), "batch1_latency_ms": round(med * 1e3, 3),
            "batch1_edit_latency_ms": round(lat5[len(lat5) // 2] * 1e3, 3),
            "stream_first_window_ms": round(lat_first[len(lat_first) // 2] * 1e3, 3)}


def workload_name(args):
    shape = (args.batch, args.phonemes, args.frames)
    if shape == (16, 50, 258) and args.precision != "bf16":
        return "C2"
    if shape == (64, 160, 861):
        return "C3" if args.precision == "bf16" else "C3 shape"
    return "custom"


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """Parent of a plain `python bench.py --gpus N`: start N children of this script, one per GPU, and wait.  The
    parent never calls into torch.cuda (a count of the GPUs may initialise the runtime, and the ranks are started from
    this process): a rank that finds no GPU under its LOCAL_RANK stops with its own message and the others are ended.
    Rank 0's stdout is the parent's, so its JSON line is the run's."""
    n = args.gpus
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 3600
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:                      # a rank failed: the others would wait at the barrier for ever
                    q.terminate()
        if time.time() > deadline:
            for q in alive:
                q.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


def claim_streams(net, device):
    """Two small batches through throughput mode: creates the model's streams and puts work on each of them."""
    from vits_amd import synthetic
    b = to_device(synthetic.config_c1(), device)
    run_steps(net, [b, b], True)
    torch.cuda.synchronize(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=258, help="frames per utterance (C2: 258; C3: 861 with --batch 64)")
    ap.add_argument("--phonemes", type=int, default=50, help="phonemes per utterance (C2: 50; C3: 160)")
    ap.add_argument("--precision", choices=("fp32", "bf16x3", "f16x2", "bf16"), default=None,
                    help="conv arithmetic: f16x2 (default), bf16x3 and fp32 are fp32-grade (the C2 parity workload); bf16 runs "
                         "the generator's convs in plain bf16 (BASELINE configs[2]: --batch 64 --frames 861 "
                         "--phonemes 160 --precision bf16)")
    ap.add_argument("--workload", choices=("batch", "c4"), default="batch",
                    help="batch (default): one fixed batch per step (C2; C3 with --batch/--frames/--phonemes); "
                         "c4: the sharded synthetic corpus of BASELINE configs[3]")
    ap.add_argument("--utterances", type=int, default=10000, help="corpus size for --workload c4")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend for the timing barrier and the max-over-ranks reduction (nccl = RCCL; "
                         "gloo lets several ranks rehearse on one GPU: ranks then share device LOCAL_RANK %% device_count)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="plain back-to-back infer calls in the timed region (default: infer_batches, which overlaps "
                         "the front end of the next batch with the generator of the current one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency section (profiling runs)")
    ap.add_argument("--no-sub-results", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--serial-generator", action="store_true",
                    help="keep the generator's ResBlock chains on one stream (rocprofv3 runs: per-kernel durations "
                         "then do not overlap and match the event-timed roofline pass)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))            # before anything in this process touches a GPU

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would report the wrong GPU count")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
    if args.backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} needs a GPU per rank, this node shows "
                         f"{torch.cuda.device_count()} (several ranks can share a GPU for a rehearsal with --backend gloo)")
    device = torch.device("cuda", local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist = None
    want_dist = world > 1 or os.environ.get("VITS_BENCH_FORCE_DIST") == "1"      # the latter: the collectives with one rank

    def init_dist():
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            td.init_process_group(backend="nccl", device_id=device)
        else:
            td.init_process_group(backend="gloo")
        return td
    early = os.environ.get("VITS_BENCH_DIST_EARLY") == "1"      # the order that costs a quarter of the throughput, kept to show it
    if want_dist and early:
        dist = init_dist()
    red_dev = device if args.backend == "nccl" else torch.device("cpu")      # where the elapsed-time reduction lives

    if args.serial_generator:
        os.environ["VITS_GEN_STREAMS"] = "1"
    from vits_amd import synthetic
    net = build_model(device, args.precision)
    if want_dist and not early:
        # A HIP process schedules four hardware queues and hands them to streams as the streams first carry work.
        # The model's front-end, generator and chain streams take theirs here, before the communicator's streams
        # exist: those are idle while steps are timed and may share a queue, the model's may not (with the group
        # set up first a C2 step took 15.1 ms instead of 11.5).
        claim_streams(net, device)
        dist = init_dist()
    if args.workload == "c4":
        run_corpus(net, args, device, rank, world, dist, red_dev)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    # every rank synthesises its own utterances: same shapes, different ids / durations / speakers / noise
    batch = synthetic.make_batch(1 + rank, [args.phonemes] * args.batch, [args.frames] * args.batch)
    d = to_device(batch, device)
    audio_s = batch.audio_seconds

    def barrier():
        torch.cuda.synchronize(device)          # this rank's work is complete before the collective runs
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    overlap = not args.no_overlap
    latency = batch1_latencies(net, device) if rank == 0 and not args.no_latency else None
    run_steps(net, [d] * args.warmup, overlap)
    barrier()
    t0 = time.perf_counter()
    run_steps(net, [d] * args.steps, overlap)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        # dominant kernel: per-launch HIP-event timing over `steps` more steps of the same workload
        roofline = roofline_pass(net, lambda: step(net, d), args.steps, device,
                                 traffic_key=dict(batch=args.batch, frames=args.frames, phonemes=args.phonemes,
                                                  precision=net.precision))
        flops_step = algorithmic_flops([args.phonemes] * args.batch, [args.frames] * args.batch)
        arith = "bf16 generator convs" if net.precision == "bf16" else f"fp32-grade {net.precision} convs"
        result = {
            "metric": "audio-sec/sec at 44.1 kHz",
            "value": round(world * args.steps * audio_s / elapsed, 2),
            "unit": "audio-s/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype_of(net),
            "conv_precision": net.precision,
            "data": "synthetic",
            "config": {"workload": f"{workload_name(args)}: batch {args.batch} x "
                                   f"{args.phonemes} phonemes x {args.frames} frames "
                                   f"({args.frames * 512 / 44100:.1f} s each), fixed durations, fixed noise, "
                                   f"{arith}; one batch per GPU per step",
                       "batch_per_gpu": args.batch, "frames": args.frames, "parallelism": f"utterance-dp{world}",
                       "front_end_overlap": overlap},
            "whole_path_tflops": round(world * args.steps * flops_step / elapsed / 1e12, 2),
            "roofline": roofline,
        }
        if latency is not None:
            # BASELINE's metric has a second half, batch-1 RTF: it rides in `config` and, identically, at the top level
            result["config"].update(latency)
            result["roofline"].update(rtf_batch1=latency["rtf_batch1"])
            result.update(latency)
        if world == 1 and not args.no_cpu_baseline:
            # about 10-30 s of CPU work: the whole 16-utterance batch at 3 s an utterance, four utterances of the long ones
            cb, cpu_out, cpu_batch = cpu_baseline(args.phonemes, args.frames,
                                                  sample_utts=min(args.batch, 16 if args.frames <= 300 else 4))
            # the same sample through the GPU path: keeps the two numbers about the same work, and checks parity
            dd = to_device(cpu_batch, device)
            gpu_out = step(net, dd).cpu()
            cb["gpu_vs_oracle_max_abs"] = float((gpu_out - cpu_out).abs().max())
            result["cpu_baseline"] = cb
        default_line = (workload_name(args) == "C2" and net.precision == "f16x2" and world == 1)
        if default_line and not args.no_sub_results:
            result["sub_results"] = sub_results(net, device, overlap)
        result["range_overflow"] = net.range_overflow()
        line = json.dumps(result)
        print(f"result line: {len(line)} bytes", file=sys.stderr, flush=True)
        print(line, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
