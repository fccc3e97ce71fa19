"""Packed weight file (vits_amd/packed.py): what the one-time conversion writes is exactly what the load-time
preparation produces from the checkpoint, the mapping refuses damaged files, and (on the GPU) a model built from the
file synthesises the same bytes as one built from the state dict."""
import json
import os

import pytest
import torch

from vits_amd import _native as N
from vits_amd import packed, synthetic, weights


@pytest.fixture(scope="module")
def packed_file(synthetic_sd, tmp_path_factory):
    path = str(tmp_path_factory.mktemp("packed") / "model.vitsw")
    header = packed.save_packed(path, synthetic_sd, synthetic.ModelConfig(), "f16x2")
    return path, header


def test_roundtrip_equals_prepare(synthetic_sd, packed_file):
    path, header = packed_file
    cfg, precision, pw = packed.load_packed(path)
    assert cfg == synthetic.ModelConfig() and precision == "f16x2"
    assert os.path.getsize(path) % 256 == 0 and all(r["offset"] % 256 == 0 for r in header["tensors"])
    ref = weights.prepare(synthetic_sd, cfg, N.PREC_F16X2)
    assert set(ref.plain) == set(pw.plain) and set(ref.convs) == set(pw.convs)
    for k, t in ref.plain.items():
        assert torch.equal(t, pw.plain[k]), k
    for k, a in ref.convs.items():
        b = pw.convs[k]
        assert torch.equal(a[0], b[0]) and a[2:] == b[2:], k
        assert (a[1] is None) == (b[1] is None) and (a[1] is None or torch.equal(a[1], b[1])), k


def test_damaged_files_are_refused(packed_file, tmp_path):
    path, _ = packed_file
    raw = open(path, "rb").read()
    bad = tmp_path / "bad.vitsw"
    bad.write_bytes(b"NOTVITS!" + raw[8:])
    with pytest.raises(ValueError, match="not a vits_amd packed"):
        packed.load_packed(str(bad))
    bad.write_bytes(raw[:len(raw) // 2])
    with pytest.raises(ValueError, match="truncated"):
        packed.load_packed(str(bad))
    bad.write_bytes(raw[:12])
    with pytest.raises(ValueError):
        packed.load_packed(str(bad))


def test_conversion_validates_the_checkpoint(synthetic_sd, tmp_path):
    sd = dict(synthetic_sd)
    del sd["dec.conv_post.weight"]
    with pytest.raises(KeyError, match="dec.conv_post.weight"):
        packed.save_packed(str(tmp_path / "x.vitsw"), sd)
    sd = dict(synthetic_sd)
    sd["emb_g.weight"] = sd["emb_g.weight"][:, :100]
    with pytest.raises(ValueError, match="size mismatch for emb_g.weight"):
        packed.save_packed(str(tmp_path / "x.vitsw"), sd)
    assert not os.path.exists(tmp_path / "x.vitsw")


def test_command_line_converter(synthetic_sd, tmp_path):
    """`python -m vits_amd.packed ckpt config out`: reads the reference's checkpoint dict (utils.py:65-68) with
    weights_only=True and a config in the reference's format; n_vocab comes from the checkpoint's symbol table."""
    cfg = synthetic.ModelConfig()
    ck = tmp_path / "G_1.pth"
    torch.save({"model": synthetic_sd, "iteration": 1, "optimizer": {}, "learning_rate": 2e-4}, str(ck))
    conf = {"train": {"segment_size": cfg.segment_size * cfg.hop_length},
            "data": {"filter_length": 2048, "hop_length": 512, "sampling_rate": 44100, "n_speakers": cfg.n_speakers},
            "model": {k: v for k, v in cfg.kwargs().items() if k != "n_speakers"}}
    cj = tmp_path / "config.json"
    cj.write_text(json.dumps(conf))
    out = tmp_path / "m.vitsw"
    packed.main([str(ck), str(cj), str(out), "--precision", "bf16x3"])
    got_cfg, precision, pw = packed.load_packed(str(out))
    assert got_cfg == cfg and precision == "bf16x3"
    assert {v[7] for v in pw.convs.values()} == {N.PREC_BF16X3}


@pytest.mark.gpu
def test_model_from_packed_matches_state_dict(synthetic_sd, packed_file):
    from vits_amd import SynthesizerTrn
    path, _ = packed_file
    cfg = synthetic.ModelConfig()
    a = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                       **cfg.kwargs()).eval()
    a.load_state_dict(synthetic_sd)
    b = SynthesizerTrn.from_packed(path)
    assert b.precision == "f16x2" and b.cfg == cfg
    x = synthetic.make_batch(21, [15, 11], [70, 52])
    kw = dict(sid=x.sid, noise_scale=x.noise_scale, duration_control=x.durations, noise=x.noise)
    oa, ob = a.infer(x.phonemes, x.lengths, **kw), b.infer(x.phonemes, x.lengths, **kw)
    assert torch.equal(oa[0], ob[0]) and torch.equal(oa[2][0], ob[2][0])
