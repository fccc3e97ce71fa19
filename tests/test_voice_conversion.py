"""voice_conversion (reference models.py:769-777): posterior encoder -> flow forward (source speaker) -> flow reverse
(target speaker) -> generator.  tests/golden/vc.npz holds the reference's outputs on a ragged batch of synthetic
spectrograms (tools/make_goldens.py: torch.manual_seed right before the call pins the posterior noise)."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN
from oracle import metrics
from vits_amd import synthetic

TOL = 2e-5           # oracle against reference (measured 1.5e-6 wav, 3.6e-6 latents)
WAV_TOL = 1e-4       # GPU against reference: the fp32 tolerances of the synthesis path
MEL_TOL = 1e-3
LATENT_TOL = 2e-4


def load_vc():
    d = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "vc.npz")).items()}
    y, y_lengths = synthetic.make_spectrogram(31, [37, 29])
    return d, y, y_lengths


def test_oracle_voice_conversion_matches_reference(oracle_model):
    d, y, y_lengths = load_vc()
    o, mask, (z, z_p, z_hat) = oracle_model.voice_conversion(y, y_lengths, d["in_sid_src"], d["in_sid_tgt"],
                                                             noise=d["in_noise"])
    assert torch.equal(mask, d["out_y_mask"]) and o.shape == d["out_o_hat"].shape == (2, 1, 37 * 512)
    assert metrics.max_abs(o, d["out_o_hat"]) <= TOL
    for got, key in ((z, "out_z"), (z_p, "out_z_p"), (z_hat, "out_z_hat")):
        assert metrics.max_abs(got, d[key]) <= TOL, key
    # the conversion does something: z_hat differs from z, and the padded tail of the latents is zero
    assert metrics.max_abs(z_hat, z) > 0.1 and float(z[1, :, 29:].abs().max()) == 0.0


def test_flow_forward_then_reverse_is_identity(oracle_model):
    """Same speaker both ways: the coupling block is invertible (a size-independent property of the path)."""
    torch.manual_seed(3)
    z = torch.randn(2, 192, 50)
    mf = (torch.arange(50)[None, :] < torch.tensor([50, 33])[:, None]).float().unsqueeze(1)
    g = oracle_model.w["emb_g.weight"][torch.tensor([7, 9])].unsqueeze(-1)
    with torch.no_grad():
        back = oracle_model.flow_reverse(oracle_model.flow_forward(z * mf, mf, g), mf, g)
    assert metrics.max_abs(back, z * mf) <= 1e-5


@pytest.mark.gpu
def test_gpu_voice_conversion(synthetic_sd, oracle_model):
    from vits_amd import SynthesizerTrn
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         **cfg.kwargs()).eval()
    net.load_state_dict(synthetic_sd)
    d, y, y_lengths = load_vc()
    o, mask, (z, z_p, z_hat) = net.voice_conversion(y, y_lengths, d["in_sid_src"], d["in_sid_tgt"], noise=d["in_noise"])
    assert o.is_cuda and mask.dtype == torch.float32 and torch.equal(mask.cpu(), d["out_y_mask"])
    for got, key in ((z, "out_z"), (z_p, "out_z_p"), (z_hat, "out_z_hat")):
        assert metrics.max_abs(got, d[key]) <= LATENT_TOL, key
    assert metrics.max_abs(o, d["out_o_hat"]) <= WAV_TOL
    assert metrics.log_mel_l1(o[:, 0], d["out_o_hat"][:, 0]) <= MEL_TOL
    # longer, equal-length batch against the oracle; and forward + reverse with one speaker returns z
    y2, l2 = synthetic.make_spectrogram(32, [120, 120, 120])
    src, tgt = torch.tensor([1, 2, 3]), torch.tensor([4, 5, 1])
    noise = torch.randn(3, 192, 120, generator=torch.Generator().manual_seed(9))
    o2, _, (za, zb, zc) = net.voice_conversion(y2, l2, src, tgt, noise=noise)
    ro, _, (rza, rzb, rzc) = oracle_model.voice_conversion(y2, l2, src, tgt, noise=noise)
    assert metrics.max_abs(o2, ro) <= WAV_TOL and metrics.max_abs(zc, rzc) <= LATENT_TOL
    same = net.voice_conversion(y2, l2, src, src, noise=noise)[2]
    assert metrics.max_abs(same[2], same[0]) <= 1e-4
    # random-noise path and error behaviour
    assert torch.isfinite(net.voice_conversion(y, y_lengths, d["in_sid_src"], d["in_sid_tgt"])[0]).all()
    with pytest.raises(ValueError):
        net.voice_conversion(y[:, :100], y_lengths, d["in_sid_src"], d["in_sid_tgt"])
    with pytest.raises(IndexError):
        net.voice_conversion(y, y_lengths, torch.tensor([5, 4000]), d["in_sid_tgt"])
    no_q = {k: v for k, v in synthetic_sd.items() if not k.startswith("enc_q.")}
    net2 = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                          **cfg.kwargs()).eval()
    net2.load_state_dict(no_q)
    with pytest.raises(RuntimeError, match="enc_q"):
        net2.voice_conversion(y, y_lengths, d["in_sid_src"], d["in_sid_tgt"])
