#!/usr/bin/env python3
"""Generate golden vectors from the reference implementation (build container only).

Run as:  PYTORCH_JIT=0 PYTHONDONTWRITEBYTECODE=1 python3 tools/make_goldens.py

Imports the reference from /root/reference (read-only), loads the deterministic synthetic checkpoint from
vits_amd.synthetic with load_state_dict(strict=True), runs SynthesizerTrn.infer on the fixture cases with a
fixed noise tensor, and writes small .npz fixtures under tests/golden/.  Only data is written: inputs, the
noise, outputs and stage-boundary tensors.  The reference itself never leaves this container.

PYTORCH_JIT=0 is required because the reference's commons.py carries GBK-encoded comments that
torch.jit.script cannot re-read; with the JIT disabled the decorated function runs eagerly with identical math.
"""
import hashlib
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)

import models  # noqa: E402  (reference)
import utils   # noqa: E402  (reference)
from text.symbols import symbols  # noqa: E402  (reference)

from vits_amd import synthetic  # noqa: E402
from oracle.mvts_oracle import OracleSynthesizer  # noqa: E402
from oracle import metrics  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def load_reference(n_speakers=None, **model_overrides):
    raw = open(os.path.join(REF, "configs/config.json"), "rb").read().decode("gbk")
    raw = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", raw, flags=re.S))
    cfg_json = json.loads(raw)
    cfg_json["model"].update(model_overrides)
    if n_speakers is not None:
        cfg_json["data"]["n_speakers"] = n_speakers
    hps = utils.HParams(**cfg_json)
    net = models.SynthesizerTrn(len(symbols), hps.data.filter_length // 2 + 1, hps.data.hop_length,
                                hps.data.sampling_rate, hps.train.segment_size // hps.data.hop_length,
                                n_speakers=hps.data.n_speakers, **hps.model).eval()
    return net, hps


def run_reference(net, batch, noise_seed, stages, **kw):
    """One reference infer with torch.randn_like replaced by a seeded generator; returns outputs + noise."""
    hooks = []
    grab = {}

    def hook(name):
        def fn(mod, inp, out):
            grab[name] = out
        return fn
    for name in ("enc_p", "duration_predictor", "pitch_predictor", "energy_predictor", "lr", "frame_prior_net",
                 "project", "flow", "dec"):
        hooks.append(getattr(net, name).register_forward_hook(hook(name)))
    # enc_p's output is modified in place later (x += prenet), so clone it inside the hook
    hooks.append(net.enc_p.register_forward_hook(lambda m, i, o: grab.__setitem__("enc_p_x", o[0].clone())))
    gen = torch.Generator().manual_seed(noise_seed)
    noise_box = {}
    orig = torch.randn_like

    def fixed_randn_like(t, **_):
        noise_box["noise"] = torch.randn(t.shape, generator=gen, dtype=t.dtype)
        return noise_box["noise"]
    torch.randn_like = fixed_randn_like
    try:
        with torch.no_grad():
            out = net.infer(batch.phonemes, batch.lengths, sid=batch.sid, **kw)
    finally:
        torch.randn_like = orig
        for h in hooks:
            h.remove()
    o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy = out
    stages.update(enc_p=grab["enc_p_x"], x_frame=grab["lr"][0], x_lengths=grab["lr"][1],
                  prior=grab["frame_prior_net"].transpose(1, 2))
    if "duration_predictor" in grab:
        stages["logw"] = grab["duration_predictor"]
    if "pitch_predictor" in grab:
        stages["lf0_pred"] = grab["pitch_predictor"]
    if "energy_predictor" in grab:
        stages["energy_pred"] = grab["energy_predictor"]
    return dict(o=o, x_mask=x_mask, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p, duration=duration, f0=f0, energy=energy,
                noise=noise_box["noise"])


def to_np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def make_vc(net, oracle, report):
    """voice_conversion (reference models.py:769-777) on a ragged batch of synthetic spectrograms; the posterior
    encoder's torch.randn_like draw is pinned by seeding right before the call."""
    y, y_lengths = synthetic.make_spectrogram(31, [37, 29])
    sid_src, sid_tgt = torch.tensor([5, 40]), torch.tensor([17, 3])
    torch.manual_seed(2031)
    with torch.no_grad():
        o_hat, y_mask, (z, z_p, z_hat) = net.voice_conversion(y, y_lengths, sid_src, sid_tgt)
    torch.manual_seed(2031)
    noise = torch.randn(2, 192, 37)
    oo, omask, (oz, ozp, ozh) = oracle.voice_conversion(y, y_lengths, sid_src, sid_tgt, noise=noise)
    errs = {"o_hat": metrics.max_abs(oo, o_hat), "z": metrics.max_abs(oz, z), "z_p": metrics.max_abs(ozp, z_p),
            "z_hat": metrics.max_abs(ozh, z_hat)}
    assert torch.equal(omask, y_mask)
    report["cases"]["vc"] = dict(oracle_vs_reference_max_abs=errs, wav_std=float(o_hat.std()),
                                 wav_absmax=float(o_hat.abs().max()), frames=[37, 29])
    print("vc", json.dumps(report["cases"]["vc"]))
    np.savez_compressed(os.path.join(OUT, "vc.npz"), in_sid_src=to_np(sid_src), in_sid_tgt=to_np(sid_tgt),
                        in_noise=to_np(noise), out_o_hat=to_np(o_hat), out_y_mask=to_np(y_mask), out_z=to_np(z),
                        out_z_p=to_np(z_p), out_z_hat=to_np(z_hat))


RESBLOCK2 = dict(resblock="2", resblock_dilation_sizes=((1, 3), (1, 3), (1, 3)))


def make_resblock2(report):
    """The reference built with modules.ResBlock2 (resblock "2", two dilations per kernel size; everything else as in
    configs/config.json): one ragged batch, waveform and latents."""
    cfg = synthetic.ModelConfig(**RESBLOCK2)
    net, _ = load_reference(resblock="2", resblock_dilation_sizes=[list(d) for d in RESBLOCK2["resblock_dilation_sizes"]])
    manifest = synthetic.state_dict_manifest(cfg)
    ref_sd = net.state_dict()
    assert list(manifest.keys()) == list(ref_sd.keys()), "ResBlock2 manifest order/names differ from the reference"
    for k, shp in manifest.items():
        assert tuple(ref_sd[k].shape) == tuple(shp), (k, ref_sd[k].shape, shp)
    sd = synthetic.synthetic_state_dict(cfg)
    net.load_state_dict(sd, strict=True)
    oracle = OracleSynthesizer(sd, resblock="2", resblock_dilation_sizes=RESBLOCK2["resblock_dilation_sizes"])
    b = synthetic.make_batch(17, [13, 8], [42, 27])
    kw = dict(noise_scale=0.667, duration_control=b.durations)
    stages = {}
    ref = run_reference(net, b, 1717, stages, **kw)
    orc = oracle.infer(b.phonemes, b.lengths, sid=b.sid, noise=ref["noise"], **kw)
    errs = {"o": metrics.max_abs(orc[0], ref["o"]), "z": metrics.max_abs(orc[2][0], ref["z"])}
    report["cases"]["resblock2_b2"] = dict(oracle_vs_reference_max_abs=errs, wav_std=float(ref["o"].std()),
                                           wav_absmax=float(ref["o"].abs().max()))
    print("resblock2_b2", json.dumps(report["cases"]["resblock2_b2"]))
    np.savez_compressed(os.path.join(OUT, "resblock2_b2.npz"), in_phonemes=to_np(b.phonemes), in_lengths=to_np(b.lengths),
                        in_sid=to_np(b.sid), in_duration_control=to_np(b.durations), out_noise=to_np(ref["noise"]),
                        out_o=to_np(ref["o"]), out_z=to_np(ref["z"]), out_x_mask=to_np(ref["x_mask"]),
                        meta_json=np.frombuffer(json.dumps(dict(noise_scale=0.667)).encode(), dtype=np.uint8))


SPEAKERLESS = dict(n_speakers=0, gin_channels=0)


def make_speakerless(report):
    """The reference built without speakers (n_speakers = 0, gin_channels = 0: g = None, models.py:719-722) on the one
    kind of call its infer can run that way -- energy given as a tensor (its energy predictor needs g,
    frame_prior_network.py:125); durations and pitch predicted in one case, all three given in the other."""
    cfg = synthetic.ModelConfig(**SPEAKERLESS)
    net, _ = load_reference(n_speakers=0, gin_channels=0)
    manifest = synthetic.state_dict_manifest(cfg)
    ref_sd = net.state_dict()
    assert list(manifest.keys()) == list(ref_sd.keys()), "speaker-less manifest order/names differ from the reference"
    for k, shp in manifest.items():
        assert tuple(ref_sd[k].shape) == tuple(shp), (k, ref_sd[k].shape, shp)
    sd = synthetic.synthetic_state_dict(cfg)
    net.load_state_dict(sd, strict=True)
    oracle = OracleSynthesizer(sd, n_speakers=0)
    b = synthetic.make_batch(19, [11, 8], [38, 26], with_controls=True)
    cases = {"speakerless_given": dict(noise_scale=0.667, duration_control=b.durations, pitch_control=b.pitch_hz,
                                       energy_control=b.energy),
             "speakerless_predicted": dict(noise_scale=0.5, duration_control=1.4, pitch_control=0.9,
                                           energy_control=b.energy)}
    for name, kw in cases.items():
        stages = {}
        ref = run_reference(net, b, 1900 + len(name), stages, **kw)
        orc = oracle.infer(b.phonemes, b.lengths, sid=None, noise=ref["noise"], **kw)
        errs = {"o": metrics.max_abs(orc[0], ref["o"]), "z": metrics.max_abs(orc[2][0], ref["z"]),
                "f0": metrics.max_abs(orc[4], ref["f0"])}
        assert torch.equal(orc[1], ref["x_mask"]), name
        report["cases"][name] = dict(oracle_vs_reference_max_abs=errs, wav_std=float(ref["o"].std()),
                                     wav_absmax=float(ref["o"].abs().max()),
                                     frames=[int(v) for v in stages["x_lengths"]])
        print(name, json.dumps(report["cases"][name]))
        arrays = {"in_phonemes": b.phonemes, "in_lengths": b.lengths}
        meta = {}
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                arrays["in_" + k] = v
            else:
                meta[k] = v
        arrays.update({"out_" + k: v for k, v in ref.items()})
        arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **{k: to_np(v) for k, v in arrays.items()})
    # without an energy tensor the reference cannot run such a model: pin the exception type
    try:
        with torch.no_grad():
            net.infer(b.phonemes, b.lengths, duration_control=b.durations)
        raise AssertionError("the reference ran a speaker-less model without an energy tensor")
    except TypeError as e:
        report["cases"]["speakerless_given"]["without_energy_tensor"] = f"TypeError: {e}"[:120]


def main():
    os.makedirs(OUT, exist_ok=True)
    cfg = synthetic.ModelConfig()
    if "--speakerless-only" in sys.argv:   # add / refresh the speaker-less vectors without touching the others
        path = os.path.join(OUT, "ORACLE_VS_REFERENCE.json")
        report = json.load(open(path))
        make_speakerless(report)
        with open(path, "w") as f:
            json.dump(report, f, indent=1)
        return
    if "--resblock2-only" in sys.argv:   # add / refresh the ResBlock2 vectors without touching the others
        path = os.path.join(OUT, "ORACLE_VS_REFERENCE.json")
        report = json.load(open(path))
        make_resblock2(report)
        with open(path, "w") as f:
            json.dump(report, f, indent=1)
        return
    net, hps = load_reference()
    if "--vc-only" in sys.argv:          # add / refresh the voice-conversion vectors without touching the others
        sd = synthetic.synthetic_state_dict(cfg)
        net.load_state_dict(sd, strict=True)
        path = os.path.join(OUT, "ORACLE_VS_REFERENCE.json")
        report = json.load(open(path))
        make_vc(net, OracleSynthesizer(sd), report)
        with open(path, "w") as f:
            json.dump(report, f, indent=1)
        return
    ref_sd = net.state_dict()
    manifest = synthetic.state_dict_manifest(cfg)
    assert list(manifest.keys()) == list(ref_sd.keys()), "manifest order/names differ from the reference"
    for k, shp in manifest.items():
        assert tuple(ref_sd[k].shape) == tuple(shp), (k, ref_sd[k].shape, shp)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump({k: list(v) for k, v in manifest.items()}, f, indent=0)
    sd = synthetic.synthetic_state_dict(cfg)
    net.load_state_dict(sd, strict=True)
    oracle = OracleSynthesizer(sd)
    # digest of the synthetic checkpoint so the GPU box can check it regenerated the same bytes
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.numpy().tobytes())
    report = {"state_dict_sha256": h.hexdigest(), "cases": {}}
    # digest of the reference's symbol inventory (text/symbols.py): pins the phoneme -> id order of vits_amd/text.py
    from text.symbols import symbols as ref_symbols
    report["symbols_sha256"] = hashlib.sha256("\n".join(ref_symbols).encode()).hexdigest()
    report["n_symbols"] = len(ref_symbols)

    mk = synthetic.make_batch
    cases = {}
    # (i) B=1, P=12, T=40, all stage boundaries, durations given
    b = mk(11, [12], [40])
    cases["small_b1"] = (b, dict(noise_scale=0.667, duration_control=b.durations))
    # (ii) B=2 ragged: pins the padding-leak semantics
    b = mk(12, [14, 9], [45, 31])
    cases["ragged_b2"] = (b, dict(noise_scale=0.667, duration_control=b.durations))
    # (iii) predicted durations, scalar controls != 1
    b = mk(13, [10, 7], [1, 1])
    cases["predicted_scalar"] = (b, dict(noise_scale=0.5, duration_control=1.3, pitch_control=1.1, energy_control=0.9))
    # (iv) tensor overrides incl. F0 = 0 and duration 0, float durations with fractions, [B,1,P] shape
    b = mk(14, [12, 12], [36, 30], with_controls=True)
    d = b.durations.clone().float()
    d[0, 3] = 0.0
    d[1, 5] = 0.0
    d[0, 1] += 0.75           # int() truncation
    d[1, 2] = -2.0            # clamped to 0
    cases["tensor_controls"] = (b, dict(noise_scale=0.667, duration_control=d.unsqueeze(1),
                                        pitch_control=b.pitch_hz, energy_control=b.energy))
    # (v) max_len truncation
    b = mk(15, [12], [40])
    cases["max_len"] = (b, dict(noise_scale=0.667, duration_control=b.durations, max_len=25))
    # (vii) very short sequences (L < window + 1 exercises the relative-embedding slice, attentions.py:277-290)
    b = mk(16, [3], [4])
    cases["tiny"] = (b, dict(noise_scale=0.667, duration_control=b.durations))

    for name, (batch, kw) in cases.items():
        stages = {}
        t0 = time.time()
        ref = run_reference(net, batch, 1000 + len(name), stages, **kw)
        t_ref = time.time() - t0
        ost = {}
        orc = oracle.infer(batch.phonemes, batch.lengths, sid=batch.sid, noise=ref["noise"], stages=ost, **kw)
        errs = {
            "o": metrics.max_abs(orc[0], ref["o"]), "z": metrics.max_abs(orc[2][0], ref["z"]),
            "z_p": metrics.max_abs(orc[2][1], ref["z_p"]), "m_p": metrics.max_abs(orc[2][2], ref["m_p"]),
            "logs_p": metrics.max_abs(orc[2][3], ref["logs_p"]), "f0": metrics.max_abs(orc[4], ref["f0"]),
            "energy": metrics.max_abs(orc[5], ref["energy"]),
            "enc_p": metrics.max_abs(ost["enc_p"], stages["enc_p"]),
            "x_frame": metrics.max_abs(ost["x_frame"], stages["x_frame"]),
            "prior": metrics.max_abs(ost["prior"], stages["prior"]),
        }
        assert torch.equal(orc[1], ref["x_mask"]), name
        assert torch.equal(ost["x_lengths"], stages["x_lengths"]), name
        if not isinstance(kw.get("duration_control"), torch.Tensor):
            assert torch.equal(orc[3], ref["duration"]), name
            errs["logw"] = metrics.max_abs(ost["logw"], stages["logw"])
        report["cases"][name] = dict(oracle_vs_reference_max_abs=errs, ref_seconds=round(t_ref, 3),
                                     wav_std=float(ref["o"].std()), wav_absmax=float(ref["o"].abs().max()),
                                     frames=[int(v) for v in stages["x_lengths"]])
        print(name, json.dumps(report["cases"][name]))
        arrays = {"in_phonemes": batch.phonemes, "in_lengths": batch.lengths, "in_sid": batch.sid}
        meta = {}
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                arrays["in_" + k] = v
            else:
                meta[k] = v
        arrays.update({"out_" + k: v for k, v in ref.items()})
        arrays.update({"stage_" + k: v for k, v in stages.items()})
        arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **{k: to_np(v) for k, v in arrays.items()})

    make_vc(net, oracle, report)
    make_resblock2(report)
    make_speakerless(report)

    # (vi) full-size C1 / C2: every 97th sample, digest, log-mel summary
    for name, batch in (("c1", synthetic.config_c1()), ("c2", synthetic.config_c2())):
        kw = dict(noise_scale=0.667, duration_control=batch.durations)
        gen = torch.Generator()
        orig = torch.randn_like
        torch.randn_like = lambda t, **_: batch.noise
        try:
            t0 = time.time()
            with torch.no_grad():
                o = net.infer(batch.phonemes, batch.lengths, sid=batch.sid, **kw)[0]
            t_ref = time.time() - t0
        finally:
            torch.randn_like = orig
        t0 = time.time()
        oo = oracle.infer(batch.phonemes, batch.lengths, sid=batch.sid, noise=batch.noise, **kw)[0]
        t_or = time.time() - t0
        lm = metrics.log_mel(o[:, 0])
        rep = dict(oracle_vs_reference_max_abs={"o": metrics.max_abs(oo, o)},
                   oracle_vs_reference_logmel_l1=metrics.log_mel_l1(oo[:, 0], o[:, 0]),
                   ref_seconds=round(t_ref, 3), oracle_seconds=round(t_or, 3),
                   wav_std=float(o.std()), wav_absmax=float(o.abs().max()),
                   sha256=hashlib.sha256(o.numpy().tobytes()).hexdigest())
        report["cases"][name] = rep
        print(name, json.dumps(rep))
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), out_o_stride97=o[:, 0, ::97].numpy(),
                            out_logmel_mean=lm.mean(dim=(1, 2)).numpy(), out_logmel_every8=lm[:, ::8, ::8].numpy())
    with open(os.path.join(OUT, "ORACLE_VS_REFERENCE.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
