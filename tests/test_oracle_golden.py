"""The CPU oracle against the golden vectors captured from the reference (tools/make_goldens.py).

The reference has no tests of its own for this path; these vectors are outputs of the reference's
SynthesizerTrn.infer (models.py:717-767) on the deterministic synthetic checkpoint."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, SMALL_CASES, infer_kwargs, load_case
from oracle import metrics
from vits_amd import synthetic

TOL = 2e-5          # fp32 rounding-order noise between two CPU evaluations (measured: <= 4e-6 on z, 1.4e-6 on o)


def test_manifest_matches_reference():
    ref = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    mine = synthetic.state_dict_manifest()
    assert list(ref.keys()) == list(mine.keys())
    assert all(tuple(ref[k]) == tuple(mine[k]) for k in ref)
    assert len(mine) == 753
    assert sum(int(np.prod(s)) for s in mine.values()) == 49622659


def test_synthetic_checkpoint_digest(synthetic_sd):
    rep = json.load(open(os.path.join(GOLDEN, "ORACLE_VS_REFERENCE.json")))
    h = hashlib.sha256()
    for k, v in synthetic_sd.items():
        h.update(k.encode())
        h.update(v.numpy().tobytes())
    assert h.hexdigest() == rep["state_dict_sha256"]


@pytest.mark.parametrize("name", SMALL_CASES)
def test_oracle_matches_reference(oracle_model, name):
    d, meta = load_case(name)
    kw = infer_kwargs(d, meta)
    st = {}
    o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy = oracle_model.infer(
        d["in_phonemes"], d["in_lengths"], sid=d["in_sid"], noise=d["out_noise"], stages=st, **kw)
    assert o.shape == d["out_o"].shape
    assert x_mask.dtype == torch.bool and torch.equal(x_mask, d["out_x_mask"])
    assert torch.equal(st["x_lengths"], d["stage_x_lengths"])
    assert metrics.max_abs(o, d["out_o"]) <= TOL
    for got, key in ((z, "out_z"), (z_p, "out_z_p"), (m_p, "out_m_p"), (logs_p, "out_logs_p"),
                     (st["enc_p"], "stage_enc_p"), (st["x_frame"], "stage_x_frame"), (st["prior"], "stage_prior")):
        assert metrics.max_abs(got, d[key]) <= TOL, key
    assert metrics.max_abs(f0, d["out_f0"]) <= 1e-3            # Hz
    assert metrics.max_abs(energy, d["out_energy"]) <= 2e-4    # raw energy units (~60..100)
    assert duration.shape == d["out_duration"].shape and duration.dtype == d["out_duration"].dtype
    assert torch.equal(duration.to(torch.float64), d["out_duration"].to(torch.float64))
    assert metrics.log_mel_l1(o[:, 0], d["out_o"][:, 0]) <= 1e-4


RESBLOCK2 = dict(resblock="2", resblock_dilation_sizes=((1, 3), (1, 3), (1, 3)))


def test_oracle_matches_reference_resblock2():
    """The generator variant built from modules.ResBlock2 (reference modules.py:228-252; resblock "2"): vectors from
    the reference constructed with that option and the synthetic checkpoint of the same configuration."""
    from oracle.mvts_oracle import OracleSynthesizer
    cfg = synthetic.ModelConfig(**RESBLOCK2)
    manifest = synthetic.state_dict_manifest(cfg)
    assert "dec.resblocks.0.convs.1.weight_v" in manifest and "dec.resblocks.0.convs1.0.weight_v" not in manifest
    sd = synthetic.synthetic_state_dict(cfg, include_enc_q=False)
    orc = OracleSynthesizer(sd, resblock="2", resblock_dilation_sizes=RESBLOCK2["resblock_dilation_sizes"])
    d, meta = load_case("resblock2_b2")
    o, x_mask, (z, *_), *_ = orc.infer(d["in_phonemes"], d["in_lengths"], sid=d["in_sid"], noise=d["out_noise"],
                                       **infer_kwargs(d, meta))
    assert torch.equal(x_mask, d["out_x_mask"])
    assert metrics.max_abs(z, d["out_z"]) <= TOL and metrics.max_abs(o, d["out_o"]) <= TOL
    assert metrics.log_mel_l1(o[:, 0], d["out_o"][:, 0]) <= 1e-4


SPEAKERLESS = dict(n_speakers=0, gin_channels=0)


@pytest.mark.parametrize("name", ["speakerless_given", "speakerless_predicted"])
def test_oracle_matches_reference_speakerless(name):
    """A model without speakers (n_speakers = 0: g = None, reference models.py:719-722) on the calls the reference's
    own infer can run that way -- energy given as a tensor; vectors from the reference built with that option."""
    from oracle.mvts_oracle import OracleSynthesizer
    cfg = synthetic.ModelConfig(**SPEAKERLESS)
    manifest = synthetic.state_dict_manifest(cfg)
    assert "emb_g.weight" not in manifest and "dec.cond.weight" not in manifest
    orc = OracleSynthesizer(synthetic.synthetic_state_dict(cfg, include_enc_q=False), n_speakers=0)
    d, meta = load_case(name)
    kw = infer_kwargs(d, meta)
    o, x_mask, (z, *_), duration, f0, energy = orc.infer(d["in_phonemes"], d["in_lengths"], noise=d["out_noise"], **kw)
    assert torch.equal(x_mask, d["out_x_mask"])
    assert torch.equal(duration.to(torch.float64), d["out_duration"].to(torch.float64))
    assert metrics.max_abs(z, d["out_z"]) <= TOL and metrics.max_abs(o, d["out_o"]) <= TOL
    assert metrics.max_abs(f0, d["out_f0"]) <= 1e-3 and metrics.max_abs(energy, d["out_energy"]) <= 2e-4
    with pytest.raises(TypeError):               # the reference's energy predictor cannot run without g
        orc.infer(d["in_phonemes"], d["in_lengths"], noise=d["out_noise"], duration_control=d["out_duration"])


def test_oracle_c1_fullsize(oracle_model):
    b = synthetic.config_c1()
    z = np.load(os.path.join(GOLDEN, "c1.npz"))
    o = oracle_model.infer(b.phonemes, b.lengths, sid=b.sid, noise=b.noise, noise_scale=b.noise_scale,
                           duration_control=b.durations)[0]
    assert o.shape == (1, 1, 258 * 512)
    assert np.abs(o[:, 0, ::97].numpy() - z["out_o_stride97"]).max() <= TOL
    lm = metrics.log_mel(o[:, 0])
    assert np.abs(lm[:, ::8, ::8].numpy() - z["out_logmel_every8"]).max() <= 1e-2


def test_oracle_bf16_generator_emulation(synthetic_sd, oracle_model):
    """generator_bf16=True (the checker for the build's plain-bf16 generator mode) leaves everything upstream of z
    untouched and stays within the bf16 envelope of the fp32 reference: measured 8e-3 max-abs, 2.7e-3 log-mel L1."""
    from oracle.mvts_oracle import OracleSynthesizer
    d, meta = load_case("ragged_b2")
    kw = infer_kwargs(d, meta)
    emu = OracleSynthesizer(synthetic_sd, generator_bf16=True)
    o, _, (z, *_), *_ = emu.infer(d["in_phonemes"], d["in_lengths"], sid=d["in_sid"], noise=d["out_noise"], **kw)
    assert metrics.max_abs(z, d["out_z"]) <= TOL
    err = metrics.max_abs(o, d["out_o"])
    assert 1e-4 < err <= 3e-2                    # really rounds (not the fp32 path), and stays in the envelope
    assert metrics.log_mel_l1(o[:, 0], d["out_o"][:, 0]) <= 1e-2
