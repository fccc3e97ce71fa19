"""Deterministic synthetic checkpoints and inputs for the MVTS synthesis path.

There is no trained checkpoint to test against, so every test, fixture and benchmark uses a state dict
produced here.  The generator is a counter-based integer hash (splitmix64) keyed by the tensor's name, so the
same bytes come out on any machine and any library version; nothing depends on ``torch.manual_seed``.

Scales are "trained-like" rather than init-like, because the reference's initial values hide bugs: the
coupling layers' ``post`` convs start at zero (reference modules.py:316-318), which makes the whole flow the
identity, and the generator convs start at N(0, 0.01) (reference commons.py:6-9), which gives a near-silent
waveform.

The tensor names and shapes follow the reference checkpoint layout (reference models.py:577-667 and the
modules it instantiates; utils.py:52-68 for the file format).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch

N_SYMBOLS = 519  # len(text.symbols.symbols) in the reference (text/symbols.py:45)


@dataclass
class ModelConfig:
    """Constructor arguments of the reference SynthesizerTrn (models.py:582-606) with configs/config.json values."""
    n_vocab: int = N_SYMBOLS
    spec_channels: int = 1025
    hop_length: int = 512
    sampling_rate: int = 44100
    segment_size: int = 32
    inter_channels: int = 192
    hidden_channels: int = 192
    filter_channels: int = 768
    n_heads: int = 2
    n_layers: int = 4
    kernel_size: int = 3
    p_dropout: float = 0.1
    resblock: str = "1"
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    upsample_rates: Tuple[int, ...] = (8, 8, 4, 2)
    upsample_initial_channel: int = 512
    upsample_kernel_sizes: Tuple[int, ...] = (16, 16, 4, 4)
    n_speakers: int = 200
    gin_channels: int = 256
    # dimensions the reference hard-codes (models.py:540,642,644; frame_prior_network.py:65-67; attentions.py:11)
    pitch_layers: int = 6
    dur_filter: int = 256
    energy_filter: int = 768
    flow_layers: int = 4
    flow_kernel: int = 5
    n_flows: int = 4
    window_size: int = 4
    enc_q_layers: int = 16

    def kwargs(self) -> dict:
        """Keyword arguments for a SynthesizerTrn constructor (positional five first in the reference)."""
        return dict(
            inter_channels=self.inter_channels, hidden_channels=self.hidden_channels,
            filter_channels=self.filter_channels, n_heads=self.n_heads, n_layers=self.n_layers,
            kernel_size=self.kernel_size, p_dropout=self.p_dropout, resblock=self.resblock,
            resblock_kernel_sizes=list(self.resblock_kernel_sizes),
            resblock_dilation_sizes=[list(d) for d in self.resblock_dilation_sizes],
            upsample_rates=list(self.upsample_rates), upsample_initial_channel=self.upsample_initial_channel,
            upsample_kernel_sizes=list(self.upsample_kernel_sizes), n_speakers=self.n_speakers,
            gin_channels=self.gin_channels)


# ----------------------------------------------------------------------------------------------------------
# manifest: name -> shape, in the reference's state_dict order
# ----------------------------------------------------------------------------------------------------------

def _encoder_manifest(prefix: str, n_layers: int, c: int, f: int, k: int, heads: int, win: int):
    out = []
    dk = c // heads
    for i in range(n_layers):
        p = f"{prefix}.attn_layers.{i}"
        out += [(f"{p}.emb_rel_k", (1, 2 * win + 1, dk)), (f"{p}.emb_rel_v", (1, 2 * win + 1, dk))]
        for n in ("q", "k", "v", "o"):
            out += [(f"{p}.conv_{n}.weight", (c, c, 1)), (f"{p}.conv_{n}.bias", (c,))]
    for i in range(n_layers):
        out += [(f"{prefix}.norm_layers_1.{i}.gamma", (c,)), (f"{prefix}.norm_layers_1.{i}.beta", (c,))]
    for i in range(n_layers):
        p = f"{prefix}.ffn_layers.{i}"
        out += [(f"{p}.conv_1.weight", (f, c, k)), (f"{p}.conv_1.bias", (f,)),
                (f"{p}.conv_2.weight", (c, f, k)), (f"{p}.conv_2.bias", (c,))]
    for i in range(n_layers):
        out += [(f"{prefix}.norm_layers_2.{i}.gamma", (c,)), (f"{prefix}.norm_layers_2.{i}.beta", (c,))]
    return out


def _wn_manifest(prefix: str, hidden: int, kernel: int, n_layers: int, gin: int):
    out = []
    for i in range(n_layers):
        p = f"{prefix}.in_layers.{i}"
        out += [(f"{p}.bias", (2 * hidden,)), (f"{p}.weight_g", (2 * hidden, 1, 1)),
                (f"{p}.weight_v", (2 * hidden, hidden, kernel))]
    for i in range(n_layers):
        rs = 2 * hidden if i < n_layers - 1 else hidden
        p = f"{prefix}.res_skip_layers.{i}"
        out += [(f"{p}.bias", (rs,)), (f"{p}.weight_g", (rs, 1, 1)), (f"{p}.weight_v", (rs, hidden, 1))]
    if gin:
        p = f"{prefix}.cond_layer"
        out += [(f"{p}.bias", (2 * hidden * n_layers,)), (f"{p}.weight_g", (2 * hidden * n_layers, 1, 1)),
                (f"{p}.weight_v", (2 * hidden * n_layers, gin, 1))]
    return out


def state_dict_manifest(cfg: ModelConfig = ModelConfig()) -> "OrderedDict[str, Tuple[int, ...]]":
    """Names and shapes of every tensor in the generator checkpoint, in state_dict order."""
    c, f, k = cfg.hidden_channels, cfg.filter_channels, cfg.kernel_size
    ic, gin = cfg.inter_channels, cfg.gin_channels
    m: List[Tuple[str, Tuple[int, ...]]] = []
    # enc_p (models.py:143-181)
    m.append(("enc_p.symbol_emb.weight", (cfg.n_vocab, c)))
    m += _encoder_manifest("enc_p.encoder", cfg.n_layers, c, f, k, cfg.n_heads, cfg.window_size)
    m += [("enc_p.proj.weight", (2 * ic, c, 1)), ("enc_p.proj.bias", (2 * ic,))]
    # dec (models.py:257-283)
    ch0 = cfg.upsample_initial_channel
    m += [("dec.conv_pre.weight", (ch0, ic, 7)), ("dec.conv_pre.bias", (ch0,))]
    for i, (u, ku) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, cout = ch0 // 2 ** i, ch0 // 2 ** (i + 1)
        m += [(f"dec.ups.{i}.bias", (cout,)), (f"dec.ups.{i}.weight_g", (cin, 1, 1)),
              (f"dec.ups.{i}.weight_v", (cin, cout, ku))]
    n_k = len(cfg.resblock_kernel_sizes)
    for i in range(len(cfg.upsample_rates)):
        ch = ch0 // 2 ** (i + 1)
        for j, (rk, rd) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            r = i * n_k + j
            for grp in (("convs",) if str(cfg.resblock) == "2" else ("convs1", "convs2")):
                for q in range(len(rd)):
                    p = f"dec.resblocks.{r}.{grp}.{q}"
                    m += [(f"{p}.bias", (ch,)), (f"{p}.weight_g", (ch, 1, 1)), (f"{p}.weight_v", (ch, ch, rk))]
    m.append(("dec.conv_post.weight", (1, ch0 // 2 ** len(cfg.upsample_rates), 7)))
    if gin:
        m += [("dec.cond.weight", (ch0, gin, 1)), ("dec.cond.bias", (ch0,))]
    # enc_q (models.py:221-241, 640): present in checkpoints, never used by infer
    m += [("enc_q.pre.weight", (c, cfg.spec_channels, 1)), ("enc_q.pre.bias", (c,))]
    m += _wn_manifest("enc_q.enc", c, 5, cfg.enc_q_layers, gin)
    m += [("enc_q.proj.weight", (2 * ic, c, 1)), ("enc_q.proj.bias", (2 * ic,))]
    # flow (models.py:184-207, 642)
    for fl in range(cfg.n_flows):
        p = f"flow.flows.{2 * fl}"
        m += [(f"{p}.pre.weight", (c, ic // 2, 1)), (f"{p}.pre.bias", (c,))]
        m += _wn_manifest(f"{p}.enc", c, cfg.flow_kernel, cfg.flow_layers, gin)
        m += [(f"{p}.post.weight", (ic // 2, c, 1)), (f"{p}.post.bias", (ic // 2,))]
    # duration predictor (models.py:105-123, 644)
    df = cfg.dur_filter
    m += [("duration_predictor.conv_1.weight", (df, c, 3)), ("duration_predictor.conv_1.bias", (df,)),
          ("duration_predictor.norm_1.gamma", (df,)), ("duration_predictor.norm_1.beta", (df,)),
          ("duration_predictor.conv_2.weight", (df, df, 3)), ("duration_predictor.conv_2.bias", (df,)),
          ("duration_predictor.norm_2.gamma", (df,)), ("duration_predictor.norm_2.beta", (df,)),
          ("duration_predictor.proj.weight", (1, df, 1)), ("duration_predictor.proj.bias", (1,))]
    if gin:
        m += [("duration_predictor.cond.weight", (c, gin, 1)), ("duration_predictor.cond.bias", (c,))]
    # frame prior (models.py:466-495)
    m.append(("frame_prior_net.emb.weight", (121, c)))
    m += _encoder_manifest("frame_prior_net.fft_block", cfg.n_layers, c, f, k, cfg.n_heads, cfg.window_size)
    # pitch predictor (models.py:515-545)
    m += _encoder_manifest("pitch_predictor.pitch_net", cfg.pitch_layers, c, f, k, cfg.n_heads, cfg.window_size)
    m += [("pitch_predictor.proj_f0.weight", (1, c, 1)), ("pitch_predictor.proj_f0.bias", (1,))]
    if gin:
        m += [("pitch_predictor.cond.weight", (c, gin, 1)), ("pitch_predictor.cond.bias", (c,))]
    # energy predictor (frame_prior_network.py:58-129)
    ef = cfg.energy_filter
    p = "energy_predictor.predictor"
    m += [(f"{p}.conv_layer.conv_1.conv.weight", (ef, c, 3)), (f"{p}.conv_layer.conv_1.conv.bias", (ef,)),
          (f"{p}.conv_layer.layer_norm_1.weight", (ef,)), (f"{p}.conv_layer.layer_norm_1.bias", (ef,)),
          (f"{p}.conv_layer.conv_2.conv.weight", (ef, ef, 3)), (f"{p}.conv_layer.conv_2.conv.bias", (ef,)),
          (f"{p}.conv_layer.layer_norm_2.weight", (ef,)), (f"{p}.conv_layer.layer_norm_2.bias", (ef,)),
          (f"{p}.linear_layer.weight", (1, ef)), (f"{p}.linear_layer.bias", (1,)),
          (f"{p}.proj.weight", (c, 1)), (f"{p}.proj.bias", (c,))]
    if gin:
        m += [("energy_predictor.cond.weight", (c, gin, 1)), ("energy_predictor.cond.bias", (c,))]
    m += [("project.proj.weight", (2 * ic, c, 1)), ("project.proj.bias", (2 * ic,)),
          ("pitch_prenet.weight", (c, 1, 3)), ("pitch_prenet.bias", (c,)),
          ("energy_prenet.weight", (c, 1, 3)), ("energy_prenet.bias", (c,))]
    if cfg.n_speakers > 1:
        m.append(("emb_g.weight", (cfg.n_speakers, gin)))
    return OrderedDict(m)


# ----------------------------------------------------------------------------------------------------------
# counter-based uniform generator
# ----------------------------------------------------------------------------------------------------------

def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_uniform(name: str, n: int, salt: int = 0) -> np.ndarray:
    """n float32 values in [-1, 1), a pure function of (name, salt, index): splitmix64 of a per-name key."""
    key = np.uint64(_fnv1a64(name) ^ (salt * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF))
    with np.errstate(over="ignore"):
        z = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + key
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u24 = (z >> np.uint64(40)).astype(np.float64)          # 24 random bits, exact in float32
    return (u24 * (2.0 / (1 << 24)) - 1.0).astype(np.float32)


def _u(name: str, shape, amp: float, salt: int = 0) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(hash_uniform(name, n, salt) * np.float32(amp)).reshape(shape)


_SQ3 = math.sqrt(3.0)   # uniform(-a, a) has std a / sqrt(3)


def synthetic_state_dict(cfg: ModelConfig = ModelConfig(), salt: int = 0,
                         include_enc_q: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """A full generator state dict with trained-like magnitudes (see module docstring)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    manifest = state_dict_manifest(cfg)
    for name, shape in manifest.items():
        if name.startswith("enc_q.") and not include_enc_q:
            continue
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "weight_g":
            continue  # set from weight_v below
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        if leaf in ("gamma",) or (leaf == "weight" and "layer_norm" in name):
            t = 1.0 + _u(name, shape, 0.2, salt)
        elif leaf in ("beta",) or (leaf == "bias" and "layer_norm" in name):
            t = _u(name, shape, 0.2, salt)
        elif leaf in ("emb_rel_k", "emb_rel_v"):
            t = _u(name, shape, _SQ3 * shape[-1] ** -0.5, salt)
        elif name == "enc_p.symbol_emb.weight":
            t = _u(name, shape, _SQ3 * cfg.hidden_channels ** -0.5, salt)
        elif name in ("emb_g.weight", "frame_prior_net.emb.weight"):
            t = _u(name, shape, _SQ3, salt)
        elif name == "duration_predictor.proj.bias":
            t = torch.full(shape, 1.7) + _u(name, shape, 0.05, salt)   # exp(1.7) ~ 5.5 frames per phoneme
        elif name == "duration_predictor.proj.weight":
            t = _u(name, shape, 0.6 / math.sqrt(fan_in) * _SQ3, salt)
        elif name == "pitch_predictor.proj_f0.bias":
            t = torch.full(shape, 0.55) + _u(name, shape, 0.05, salt)  # LF0 0.55 ~ 190 Hz
        elif name == "energy_predictor.predictor.linear_layer.bias":
            t = _u(name, shape, 0.1, salt)
        elif leaf == "bias":
            t = _u(name, shape, 0.5 / math.sqrt(max(fan_in_of_bias(name, manifest), 1)), salt)
        elif name.startswith("dec.ups.") and leaf == "weight_v":
            # ConvTranspose1d weight [Cin, Cout, k]; each output sample sees Cin * k / stride taps
            i = int(name.split(".")[2])
            taps = shape[0] * shape[2] / cfg.upsample_rates[i]
            t = _u(name, shape, 1.4 / math.sqrt(taps) * _SQ3, salt)
        elif name.startswith("dec.resblocks.") and leaf == "weight_v":
            t = _u(name, shape, 0.7 / math.sqrt(fan_in) * _SQ3, salt)
        elif name == "dec.conv_post.weight":
            t = _u(name, shape, 0.26 / math.sqrt(fan_in) * _SQ3, salt)    # pre-tanh std ~0.2: no saturation
        elif name == "project.proj.weight":
            t = _u(name, shape, 0.4 / math.sqrt(fan_in) * _SQ3, salt)     # logs_p std ~0.4
        elif name == "enc_q.proj.weight":
            t = _u(name, shape, 0.15 / math.sqrt(fan_in) * _SQ3, salt)    # logs_q std ~0.3: a posterior of unit scale
        elif name.startswith("flow.") and ".post." in name and leaf == "weight":
            t = _u(name, shape, 0.05 * _SQ3, salt)
        elif ".attn_layers." in name and leaf == "weight":
            t = _u(name, shape, math.sqrt(6.0 / (shape[0] + shape[1])), salt)     # xavier-uniform
        else:
            t = _u(name, shape, 1.0 / math.sqrt(fan_in) * _SQ3, salt)
        sd[name] = t.contiguous()
    # weight_g = ||v|| * (1 +- 0.1), norm over all dims but 0 (torch.nn.utils.weight_norm default dim=0)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in manifest.items():
        if name.startswith("enc_q.") and not include_enc_q:
            continue
        if name.endswith(".weight_g"):
            v = sd[name[:-1] + "v"]
            nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(shape)
            out[name] = (nrm * (1.0 + _u(name, shape, 0.1, salt))).contiguous()
        else:
            out[name] = sd[name]
    return out


def fan_in_of_bias(name: str, manifest) -> int:
    base = name.rsplit(".", 1)[0]
    for wname in (base + ".weight", base + ".weight_v"):
        if wname in manifest:
            shp = manifest[wname]
            return int(np.prod(shp[1:])) if len(shp) > 1 else 1
    return 1


# ----------------------------------------------------------------------------------------------------------
# synthetic inputs (phoneme ids, MFA-style durations, controls, noise)
# ----------------------------------------------------------------------------------------------------------

def mfa_like_durations(gen: torch.Generator, n_phonemes: int, n_frames: int) -> torch.Tensor:
    """Positive integer durations summing to n_frames: floor(w / sum(w) * T), remainder to phoneme 0."""
    w = torch.rand(n_phonemes, generator=gen, dtype=torch.float64) + 0.2
    d = torch.floor(w / w.sum() * n_frames).to(torch.int64)
    d[0] += n_frames - int(d.sum())
    return d


@dataclass
class SynthBatch:
    phonemes: torch.Tensor          # [B, P] int64
    lengths: torch.Tensor           # [B] int64
    sid: torch.Tensor               # [B] int64
    durations: torch.Tensor         # [B, P] int64 (0 at padded phonemes)
    frames: torch.Tensor            # [B] int64, sum of durations
    noise: torch.Tensor             # [B, 192, T] float32
    noise_scale: float = 0.667
    pitch_hz: torch.Tensor = None   # [B, P] float32 or None
    energy: torch.Tensor = None     # [B, P] float32 or None

    @property
    def audio_seconds(self) -> float:
        return float(self.frames.sum()) * 512 / 44100.0


def make_batch(seed: int, phoneme_counts, frame_counts, channels: int = 192,
               with_controls: bool = False) -> SynthBatch:
    """Synthetic utterances: ids ~ U{1..518}, MFA-like durations, sid ~ U{0..66}, seeded N(0,1) noise."""
    gen = torch.Generator().manual_seed(int(seed))
    B = len(phoneme_counts)
    P = int(max(phoneme_counts))
    T = int(max(frame_counts))
    ph = torch.zeros(B, P, dtype=torch.int64)
    du = torch.zeros(B, P, dtype=torch.int64)
    for b, (p, t) in enumerate(zip(phoneme_counts, frame_counts)):
        ph[b, :p] = torch.randint(1, N_SYMBOLS, (p,), generator=gen)
        du[b, :p] = mfa_like_durations(gen, p, t)
    sid = torch.randint(0, 67, (B,), generator=gen)
    noise = torch.randn(B, channels, T, generator=gen, dtype=torch.float32)
    batch = SynthBatch(ph, torch.tensor(list(phoneme_counts), dtype=torch.int64), sid, du,
                       torch.tensor(list(frame_counts), dtype=torch.int64), noise)
    if with_controls:
        f0 = 100.0 + 350.0 * torch.rand(B, P, generator=gen)
        f0[torch.rand(B, P, generator=gen) < 0.1] = 0.0          # unvoiced phonemes
        batch.pitch_hz = f0.float()
        batch.energy = (100.0 * torch.rand(B, P, generator=gen)).float()
    return batch


def config_c1() -> SynthBatch:
    """One 50-phoneme utterance, 258 frames (132 096 samples, 3.0 s)."""
    return make_batch(0, [50], [258])


def config_c2(batch: int = 16) -> SynthBatch:
    """16 utterances of 50 phonemes / 258 frames each (no padding)."""
    return make_batch(1, [50] * batch, [258] * batch)


def config_c3(batch: int = 64) -> SynthBatch:
    """64 long utterances: 160 phonemes, 861 frames (440 832 samples, 10 s) each."""
    return make_batch(2, [160] * batch, [861] * batch)


def config_c5() -> SynthBatch:
    """C1 with per-phoneme F0 / energy / duration overrides."""
    return make_batch(0, [50], [258], with_controls=True)


def make_spectrogram(seed: int, frame_counts, channels: int = 1025):
    """Synthetic linear-magnitude spectrograms for the voice-conversion path: non-negative, spectral tilt towards
    low bins, zero beyond each row's length.  Returns (y [B, channels, T] float32, lengths [B] int64)."""
    gen = torch.Generator().manual_seed(int(seed))
    B, T = len(frame_counts), int(max(frame_counts))
    tilt = torch.exp(-torch.arange(channels, dtype=torch.float32) / 300.0)[None, :, None]
    y = torch.randn(B, channels, T, generator=gen).abs() * 3.0 * tilt
    lengths = torch.tensor(list(frame_counts), dtype=torch.int64)
    y = y * (torch.arange(T)[None, None, :] < lengths[:, None, None])
    return y, lengths


def corpus_c4(n_utts: int = 10000, seed: int = 4, max_batch: int = 64):
    """Lengths of a synthetic corpus, sorted by frames and cut into batches of equal composition for any rank count.

    T ~ clip(lognormal(median 386, sigma 0.6), 88, 1400) frames; P = max(5, round(T / 11)).
    Returns a list of (phoneme_counts, frame_counts) per batch.
    """
    gen = torch.Generator().manual_seed(seed)
    t = torch.exp(torch.randn(n_utts, generator=gen, dtype=torch.float64) * 0.6 + math.log(386.0))
    t = t.clamp(88, 1400).round().to(torch.int64)
    t, _ = torch.sort(t)
    batches = []
    for i in range(0, n_utts, max_batch):
        tt = t[i:i + max_batch].tolist()
        batches.append(([max(5, int(round(x / 11))) for x in tt], tt))
    return batches
