"""End-to-end parity of vits_amd.SynthesizerTrn.infer on the GPU.

Checked against (a) the golden vectors captured from the reference (tests/golden, tools/make_goldens.py) and
(b) the CPU oracle on the same seeded inputs.  Tolerances (fp32 path): waveform max-abs <= 1e-4 and log-mel L1
<= 1e-3, about 100x the fp32 noise floor measured between two CPU evaluations (1e-6 / 4e-7)."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, SMALL_CASES, infer_kwargs, load_case
from oracle import metrics
from vits_amd import synthetic

pytestmark = pytest.mark.gpu

WAV_TOL = 1e-4
MEL_TOL = 1e-3
LATENT_TOL = 2e-4      # z / z_p reach |20|; m_p, logs_p, stage tensors are O(1)


def _gpu_model(synthetic_sd):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from vits_amd import SynthesizerTrn
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         **cfg.kwargs()).eval()
    net.load_state_dict(synthetic_sd)
    return net


@pytest.fixture(scope="module")
def gpu_model(synthetic_sd):
    return _gpu_model(synthetic_sd)


@pytest.mark.parametrize("name", SMALL_CASES)
def test_infer_matches_reference_goldens(gpu_model, name):
    d, meta = load_case(name)
    kw = infer_kwargs(d, meta)
    st = {}
    o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy = gpu_model.infer(
        d["in_phonemes"], d["in_lengths"], sid=d["in_sid"], noise=d["out_noise"], stages=st, **kw)
    assert o.is_cuda and o.shape == d["out_o"].shape
    assert x_mask.dtype == torch.bool and torch.equal(x_mask.cpu(), d["out_x_mask"])
    assert torch.equal(st["x_lengths"].cpu(), d["stage_x_lengths"])
    for got, key in ((st["enc_p"], "stage_enc_p"), (st["x_frame"], "stage_x_frame"), (st["prior"], "stage_prior"),
                     (m_p, "out_m_p"), (logs_p, "out_logs_p"), (z_p, "out_z_p"), (z, "out_z")):
        assert metrics.max_abs(got, d[key]) <= LATENT_TOL, key
    assert metrics.max_abs(o, d["out_o"]) <= WAV_TOL
    assert metrics.log_mel_l1(o[:, 0], d["out_o"][:, 0]) <= MEL_TOL
    assert (f0.cpu() - d["out_f0"]).abs().max() <= 1e-3 * max(1.0, float(d["out_f0"].abs().max()))
    assert metrics.max_abs(energy, d["out_energy"]) <= 1e-3
    dur = duration.cpu()
    assert dur.shape == d["out_duration"].shape and dur.dtype == d["out_duration"].dtype
    assert torch.equal(dur.to(torch.float64), d["out_duration"].to(torch.float64))


def test_infer_c1_fullsize(gpu_model, oracle_model):
    b = synthetic.config_c1()
    g = np.load(os.path.join(GOLDEN, "c1.npz"))
    kw = dict(noise_scale=b.noise_scale, duration_control=b.durations)
    o = gpu_model.infer(b.phonemes, b.lengths, sid=b.sid, noise=b.noise, **kw)[0].cpu()
    assert o.shape == (1, 1, 258 * 512)
    assert np.abs(o[:, 0, ::97].numpy() - g["out_o_stride97"]).max() <= WAV_TOL          # reference samples
    ref = oracle_model.infer(b.phonemes, b.lengths, sid=b.sid, noise=b.noise, **kw)[0]
    assert metrics.max_abs(o, ref) <= WAV_TOL
    assert metrics.log_mel_l1(o[:, 0], ref[:, 0]) <= MEL_TOL


def test_infer_c2_batch16_properties(gpu_model):
    """Full-size batch (16 x 3 s): reference samples, and batch-composition independence for equal lengths
    (no padding => each row equals its own batch-1 result)."""
    b = synthetic.config_c2()
    g = np.load(os.path.join(GOLDEN, "c2.npz"))
    kw = dict(noise_scale=b.noise_scale, duration_control=b.durations)
    o = gpu_model.infer(b.phonemes, b.lengths, sid=b.sid, noise=b.noise, **kw)[0]
    assert o.shape == (16, 1, 258 * 512)
    assert np.abs(o[:, 0, ::97].cpu().numpy() - g["out_o_stride97"]).max() <= WAV_TOL
    lm = metrics.log_mel(o[:, 0].cpu())
    assert np.abs(lm[:, ::8, ::8].numpy() - g["out_logmel_every8"]).max() <= 5e-2
    i = 5
    o1 = gpu_model.infer(b.phonemes[i:i + 1], b.lengths[i:i + 1], sid=b.sid[i:i + 1], noise=b.noise[i:i + 1],
                         noise_scale=b.noise_scale, duration_control=b.durations[i:i + 1])[0]
    assert metrics.max_abs(o1[0], o[i]) <= 1e-5
    # determinism: same inputs, same bytes
    o2 = gpu_model.infer(b.phonemes, b.lengths, sid=b.sid, noise=b.noise, **kw)[0]
    assert torch.equal(o, o2)


def test_infer_errors(gpu_model):
    b = synthetic.make_batch(3, [6], [20])
    with pytest.raises(TypeError):
        gpu_model.infer(b.phonemes, b.lengths, duration_control=b.durations)
    bad = b.phonemes.clone()
    bad[0, 0] = 100000
    with pytest.raises(IndexError):
        gpu_model.infer(bad, b.lengths, sid=b.sid, duration_control=b.durations)
    with pytest.raises(ValueError):
        gpu_model.infer(b.phonemes, b.lengths, sid=b.sid, duration_control=torch.zeros_like(b.durations))
    # random noise path runs and has the right shapes
    o, x_mask, (z, z_p, m_p, logs_p), dur, f0, en = gpu_model.infer(b.phonemes, b.lengths, sid=b.sid, noise_scale=0.667,
                                                                    duration_control=b.durations)
    assert o.shape == (1, 1, 20 * 512) and torch.isfinite(o).all()
    assert m_p.shape == logs_p.shape == z.shape == (1, 192, 20)


def test_infer_c3_fullsize_properties(gpu_model):
    """BASELINE configs[2] shape (64 utterances x 861 frames, 160 phonemes): too large for the CPU oracle in a test,
    so it is checked through properties that do not depend on size: equal lengths => a row equals its own batch-1
    result; same inputs => same bytes; output finite and inside tanh's range."""
    b = synthetic.config_c3()
    kw = dict(noise_scale=b.noise_scale, duration_control=b.durations)
    o = gpu_model.infer(b.phonemes, b.lengths, sid=b.sid, noise=b.noise, **kw)[0]
    assert o.shape == (64, 1, 861 * 512)
    assert torch.isfinite(o).all() and float(o.abs().max()) <= 1.0
    for i in (0, 37, 63):
        o1 = gpu_model.infer(b.phonemes[i:i + 1], b.lengths[i:i + 1], sid=b.sid[i:i + 1], noise=b.noise[i:i + 1],
                             noise_scale=b.noise_scale, duration_control=b.durations[i:i + 1])[0]
        assert metrics.max_abs(o1[0], o[i]) <= 1e-5
    o2 = gpu_model.infer(b.phonemes, b.lengths, sid=b.sid, noise=b.noise, **kw)[0]
    assert torch.equal(o, o2)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x2"])
def test_infer_fp32_grade_modes(synthetic_sd, precision):
    """Every fp32-grade conv mode end to end against the reference goldens at the fp32 tolerances: "fp32"
    (v_mfma_f32_32x32x2_f32), "bf16x3" (three-term bf16 splits, six products) and "f16x2" (two-term fp16 splits,
    three products in the generator; the default, so every other test in this file runs it too)."""
    from vits_amd import SynthesizerTrn
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         precision=precision, **cfg.kwargs()).eval()
    assert net.precision == precision
    net.load_state_dict(synthetic_sd)
    for name in ("ragged_b2", "tensor_controls"):
        d, meta = load_case(name)
        o, _, (z, z_p, m_p, logs_p), *_ = net.infer(d["in_phonemes"], d["in_lengths"], sid=d["in_sid"],
                                                    noise=d["out_noise"], **infer_kwargs(d, meta))
        assert metrics.max_abs(z, d["out_z"]) <= LATENT_TOL
        assert metrics.max_abs(o, d["out_o"]) <= WAV_TOL
        assert metrics.log_mel_l1(o[:, 0], d["out_o"][:, 0]) <= MEL_TOL


BF16_WAV_TOL = 1.6e-2    # plain-bf16 generator against fp32: twice the measured 8e-3 max-abs (rms 1.5e-3, 42 dB below the signal)
BF16_MEL_TOL = 5e-3      # twice the measured 2.7e-3
BF16_RMS_TOL = 3e-3      # against the oracle emulating the same rounding: measured 1e-3


def test_infer_bf16_generator_mode(synthetic_sd):
    """precision="bf16": the generator's convs round both operands to bf16 (one MFMA product, fp32 accumulate);
    everything upstream of z stays fp32-grade, so latents, masks and durations must still meet the fp32 tolerances.

    The waveform is checked against the reference goldens (fp32) at the bf16 tolerances above, and against the oracle
    emulating the same rounding.  That emulation is not a tighter check end to end: a value that sits next to a bf16
    rounding boundary flips with any 1e-6 relative difference upstream (accumulation order), and 24 convs in series
    amplify the flips -- perturbing the oracle's own z by 1e-6 relative moves its bf16 waveform by 5e-3 max-abs /
    5e-4 rms.  The kernel's arithmetic itself is pinned per operator (tests/test_gpu_ops.py, bf16 cases: 2e-5 against
    fp32 math on rounded operands)."""
    from oracle.mvts_oracle import OracleSynthesizer
    from vits_amd import SynthesizerTrn
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         precision="bf16", **cfg.kwargs()).eval()
    net.load_state_dict(synthetic_sd)
    emu = OracleSynthesizer(synthetic_sd, generator_bf16=True)
    for name in ("ragged_b2", "predicted_scalar"):
        d, meta = load_case(name)
        kw = infer_kwargs(d, meta)
        o, x_mask, (z, z_p, m_p, logs_p), duration, *_ = net.infer(d["in_phonemes"], d["in_lengths"], sid=d["in_sid"],
                                                                   noise=d["out_noise"], **kw)
        assert torch.equal(x_mask.cpu(), d["out_x_mask"])
        assert torch.equal(duration.cpu().to(torch.float64), d["out_duration"].to(torch.float64))
        assert metrics.max_abs(z, d["out_z"]) <= LATENT_TOL
        assert o.shape == d["out_o"].shape
        assert metrics.max_abs(o, d["out_o"]) <= BF16_WAV_TOL
        assert metrics.log_mel_l1(o[:, 0], d["out_o"][:, 0]) <= BF16_MEL_TOL
        ref = emu.infer(d["in_phonemes"], d["in_lengths"], sid=d["in_sid"], noise=d["out_noise"], **kw)[0]
        assert metrics.max_abs(o, ref) <= BF16_WAV_TOL
        assert float((o.cpu() - ref).pow(2).mean().sqrt()) <= BF16_RMS_TOL
        assert metrics.log_mel_l1(o[:, 0], ref[:, 0]) <= BF16_MEL_TOL


STREAM_TOL = 1e-5        # windows against the whole utterance: same operands, kernel builds that differ in summation order


@pytest.mark.parametrize("chunk", [64, 50, 7])
def test_infer_stream_concatenates_to_infer(gpu_model, chunk):
    """Chunked synthesis (windows of `chunk` frames + the generator's receptive field as context) must reproduce the
    whole-utterance waveform: every output sample sees the same operands.  A window is a smaller grid than the whole
    utterance, and the conv launcher picks its kernel build by grid size (the 7- / 11-tap LDS-DMA build sums products 32
    at a time, the small-grid build 16 at a time), so the two agree to fp32 summation order -- STREAM_TOL, a tenth of the
    parity tolerance against the reference -- and bit for bit wherever both sides run the same builds (the latents)."""
    b = synthetic.make_batch(11, [20, 14], [150, 117])            # ragged: the padded tail is part of the contract
    kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
    full = gpu_model.infer(b.phonemes, b.lengths, **kw)
    st = gpu_model.infer_stream(b.phonemes, b.lengths, chunk_frames=chunk, **kw)
    assert st.halo_frames == 13 and st.total_frames == 150
    pieces, pos = [], 0
    for start, wav in st:
        assert start == pos and wav.shape[:2] == (2, 1)
        pieces.append(wav.clone())
        pos += wav.shape[-1]
    assert pos == st.total_samples == full[0].shape[-1]
    assert metrics.max_abs(torch.cat(pieces, -1), full[0]) <= STREAM_TOL
    assert torch.equal(st.outputs[0], full[1]) and torch.equal(st.outputs[1][0], full[2][0])


def test_infer_stream_max_len_and_single_window(gpu_model):
    b = synthetic.make_batch(12, [9], [40])
    kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
    full = gpu_model.infer(b.phonemes, b.lengths, max_len=33, **kw)[0]
    got = torch.cat([w for _, w in gpu_model.infer_stream(b.phonemes, b.lengths, max_len=33, chunk_frames=16, **kw)], -1)
    assert got.shape == full.shape == (1, 1, 33 * 512) and metrics.max_abs(got, full) <= STREAM_TOL
    one = list(gpu_model.infer_stream(b.phonemes, b.lengths, chunk_frames=1000, **kw))
    assert len(one) == 1 and torch.equal(one[0][1], gpu_model.infer(b.phonemes, b.lengths, **kw)[0])
    with pytest.raises(ValueError):
        gpu_model.infer_stream(b.phonemes, b.lengths, chunk_frames=0, **kw)


def test_infer_batches_equals_infer(gpu_model):
    """Throughput mode (front end of batch i+1 beside the generator of batch i on two streams): same bytes as
    back-to-back infer calls, for batches of different shapes, predicted and given durations, max_len."""
    bs = [synthetic.make_batch(31, [20, 14], [150, 117]), synthetic.make_batch(32, [9], [40]),
          synthetic.make_batch(33, [30, 30, 30], [200, 200, 200]), synthetic.make_batch(34, [12, 5], [1, 1])]
    kws = []
    for i, b in enumerate(bs):
        kw = dict(phonemes=b.phonemes, phonemes_lengths=b.lengths, sid=b.sid, noise_scale=b.noise_scale)
        if i == 3:
            kw.update(duration_control=1.2)                       # predicted durations: noise must be drawn inside
        else:
            kw.update(duration_control=b.durations, noise=b.noise)
        if i == 1:
            kw["max_len"] = 33
        kws.append(kw)
    torch.manual_seed(77)
    ref = []
    for kw in kws:
        k = dict(kw)
        ref.append(gpu_model.infer(k.pop("phonemes"), k.pop("phonemes_lengths"), **k))
    for _ in range(2):                                            # second pass: streams and workspaces already exist
        torch.manual_seed(77)
        got = list(gpu_model.infer_batches(kws))
        assert len(got) == len(ref)
        for g, r in zip(got, ref):
            assert torch.equal(g[0], r[0]) and torch.equal(g[1], r[1]) and torch.equal(g[2][0], r[2][0])
            assert torch.equal(g[4], r[4])
    # from a stream of the caller's own (the generator then keeps to one stream: the process's hardware queues are
    # counted), with the caller copying every batch to the host while the next one is in flight: the same bytes
    own = torch.cuda.Stream()
    with torch.cuda.stream(own):
        torch.manual_seed(77)
        got = [tuple(t.to("cpu", non_blocking=True) for t in (g[0], g[1], g[2][0], g[4]))
               for g in gpu_model.infer_batches(kws)]
        own.synchronize()
    for g, r in zip(got, ref):
        assert all(torch.equal(a, b.cpu()) for a, b in zip(g, (r[0], r[1], r[2][0], r[4])))
    assert list(gpu_model.infer_batches([])) == []
    with pytest.raises(TypeError):
        list(gpu_model.infer_batches([dict(kws[0], bogus=1)]))


def test_infer_batches_is_lazy(gpu_model):
    """infer_batches takes its input as it comes: element i + 1 is asked for only after result i - 1 was handed over,
    and a source that has nothing more for now says None, upon which the results still in the pipeline come out
    before the source is asked again -- a serving loop that builds request i + 1 only after it has delivered result i
    (reference inference_api.py:38-60 serves one request at a time).  Host tensors go up through pinned memory on the
    front-end stream.  Same bytes as plain infer."""
    bs = [synthetic.make_batch(51 + i, [10 + i, 7], [60 + 9 * i, 41]) for i in range(5)]
    kws = [dict(phonemes=b.phonemes, phonemes_lengths=b.lengths, sid=b.sid, noise_scale=b.noise_scale,
                duration_control=b.durations, noise=b.noise) for b in bs]
    ref = []
    for kw in kws:
        k = dict(kw)
        ref.append(gpu_model.infer(k.pop("phonemes"), k.pop("phonemes_lengths"), **k)[0].cpu())
    consumed, asked = [], []

    def strict_source():                       # hands out request i only after result i - 1 was consumed
        for i, kw in enumerate(kws):
            if len(consumed) < i:
                yield None                     # "nothing right now": the pipeline must drain
            assert len(consumed) == i, "a result was held back although the source had nothing to offer"
            asked.append((i, len(consumed)))
            yield kw

    for g in gpu_model.infer_batches(strict_source()):
        consumed.append(g[0].cpu())
    assert len(consumed) == len(ref) and all(torch.equal(a, b) for a, b in zip(consumed, ref))

    consumed.clear()
    asked.clear()

    def eager_source():                        # always has the next request: one element of look-ahead, no more
        for i, kw in enumerate(kws):
            asked.append((i, len(consumed)))
            yield kw

    for g in gpu_model.infer_batches(eager_source()):
        consumed.append(g[0].cpu())
    assert all(torch.equal(a, b) for a, b in zip(consumed, ref))
    assert all(i - done <= 2 for i, done in asked), asked          # request i is taken with result i - 2 delivered
    # the two modes share the model: plain infer calls straight after throughput mode, a host copy after each, keep to
    # the hardware-queue budget (the generator's chains get fewer streams once the two pipeline streams exist)
    for kw, r in zip(kws[:3], ref):
        k = dict(kw)
        assert torch.equal(gpu_model.infer(k.pop("phonemes"), k.pop("phonemes_lengths"), **k)[0].cpu(), r)
    assert gpu_model._budget_generator_streams(torch.cuda.current_stream()) == 2
    own = torch.cuda.Stream()
    with torch.cuda.stream(own):
        assert gpu_model._budget_generator_streams(own) == 1
        k = dict(kws[0])
        assert torch.equal(gpu_model.infer(k.pop("phonemes"), k.pop("phonemes_lengths"), **k)[0].cpu(), ref[0])


def test_infer_at_the_length_limits(gpu_model, oracle_model):
    """A 2048-frame utterance (the longest row the lanes-on-keys attention kernel could hold; rows of 512 and more now
    run on the matrix-core kernel) from 400 phonemes, against the oracle; a row beyond the remaining limit of 16384
    frames is refused with a clear error rather than computed wrongly."""
    b = synthetic.make_batch(41, [400], [2048])
    kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
    o = gpu_model.infer(b.phonemes, b.lengths, **kw)[0]
    assert o.shape == (1, 1, 2048 * 512)
    ref = oracle_model.infer(b.phonemes, b.lengths, **kw)[0]
    assert metrics.max_abs(o, ref) <= WAV_TOL
    assert metrics.log_mel_l1(o[:, 0], ref[:, 0]) <= MEL_TOL
    longer = synthetic.make_batch(42, [400], [16385])
    with pytest.raises(Exception, match="16384"):
        gpu_model.infer(longer.phonemes, longer.lengths, sid=longer.sid, noise_scale=0.667,
                        duration_control=longer.durations, noise=longer.noise)
    # the model still works after the refused call
    again = gpu_model.infer(b.phonemes, b.lengths, **kw)[0]
    assert torch.equal(again, o)


def _model(synthetic_sd, precision=None, cfg=None, **kw):
    from vits_amd import SynthesizerTrn
    cfg = cfg or synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         precision=precision, **cfg.kwargs(), **kw).eval()
    net.load_state_dict(synthetic_sd)
    return net


def test_infer_c3_bf16_fullsize_properties(synthetic_sd, gpu_model):
    """BASELINE configs[2] as quoted: 64 utterances x 861 frames (160 phonemes) with bf16 generator convs.  Too large
    for the CPU oracle, so: same inputs => same bytes; latents (everything upstream of the generator stays
    fp32-grade) equal to the default mode's; waveform inside the bf16 envelope (BF16_WAV_TOL / BF16_MEL_TOL, the
    tolerances test_infer_bf16_generator_mode pins against the reference goldens) of the fp32-grade waveform of the
    same batch; a row inside the same envelope of its own batch-1 result (equal lengths => no padding)."""
    net = _model(synthetic_sd, "bf16")
    b = synthetic.config_c3()
    kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
    o, x_mask, (z, *_), *_ = net.infer(b.phonemes, b.lengths, **kw)
    assert o.shape == (64, 1, 861 * 512) and bool(x_mask.all())
    assert torch.isfinite(o).all() and float(o.abs().max()) <= 1.0
    o2 = net.infer(b.phonemes, b.lengths, **kw)[0]
    assert torch.equal(o, o2)
    del o2
    ref, _, (z_ref, *_), *_ = gpu_model.infer(b.phonemes, b.lengths, **kw)          # f16x2: fp32-grade
    assert metrics.max_abs(z, z_ref) <= 1e-5
    assert metrics.max_abs(o, ref) <= BF16_WAV_TOL
    for i in (0, 41):
        assert metrics.log_mel_l1(o[i], ref[i]) <= BF16_MEL_TOL
    del ref
    i = 29
    o1 = net.infer(b.phonemes[i:i + 1], b.lengths[i:i + 1], sid=b.sid[i:i + 1], noise=b.noise[i:i + 1],
                   noise_scale=b.noise_scale, duration_control=b.durations[i:i + 1])[0]
    assert metrics.max_abs(o1[0], o[i]) <= BF16_WAV_TOL
    assert not net.range_overflow() and not gpu_model.range_overflow()


def test_infer_c3_bf16_row_against_the_oracle(synthetic_sd):
    """One row of BASELINE configs[2] (batch 1, 160 phonemes, 861 frames = 10 s) with bf16 generator convs against the
    CPU oracle -- once emulating the same operand rounding (rms bound: single values next to a bf16 rounding boundary
    flip with the accumulation order, so max-abs is held to the envelope only) and once in fp32 (the envelope) -- so
    that the full-size bf16 configuration is pinned by something other than this build's own fp32-grade output."""
    from oracle.mvts_oracle import OracleSynthesizer
    net = _model(synthetic_sd, "bf16")
    c3 = synthetic.config_c3()
    i = 17
    args = (c3.phonemes[i:i + 1], c3.lengths[i:i + 1])
    kw = dict(sid=c3.sid[i:i + 1], noise_scale=c3.noise_scale, duration_control=c3.durations[i:i + 1],
              noise=c3.noise[i:i + 1])
    o, _, (z, *_), *_ = net.infer(*args, **kw)
    assert o.shape == (1, 1, 861 * 512)
    emu_o, _, (emu_z, *_), *_ = OracleSynthesizer(synthetic_sd, generator_bf16=True).infer(*args, **kw)
    assert metrics.max_abs(z, emu_z) <= LATENT_TOL
    assert float((o.cpu() - emu_o).pow(2).mean().sqrt()) <= BF16_RMS_TOL
    assert metrics.max_abs(o, emu_o) <= BF16_WAV_TOL
    ref = OracleSynthesizer(synthetic_sd).infer(*args, **kw)[0]
    assert metrics.max_abs(o, ref) <= BF16_WAV_TOL
    assert metrics.log_mel_l1(o[:, 0], ref[:, 0]) <= BF16_MEL_TOL


@pytest.mark.parametrize("name", ["speakerless_given", "speakerless_predicted"])
def test_infer_speakerless_matches_reference(name):
    """n_speakers = 0 (reference models.py:719-722: g = None, no conditioning conv applied anywhere), on the calls the
    reference's own infer can run that way -- energy_control a tensor; without one it raises TypeError
    (frame_prior_network.py:125), and so does this build.  Vectors from the reference built with n_speakers = 0,
    gin_channels = 0; sid is ignored, as there."""
    cfg = synthetic.ModelConfig(n_speakers=0, gin_channels=0)
    net = _model(synthetic.synthetic_state_dict(cfg, include_enc_q=False), cfg=cfg)
    d, meta = load_case(name)
    kw = infer_kwargs(d, meta)
    o, x_mask, (z, z_p, m_p, logs_p), duration, f0, energy = net.infer(d["in_phonemes"], d["in_lengths"],
                                                                       noise=d["out_noise"], **kw)
    assert torch.equal(x_mask.cpu(), d["out_x_mask"])
    assert torch.equal(duration.cpu().to(torch.float64), d["out_duration"].to(torch.float64))
    for got, key in ((m_p, "out_m_p"), (logs_p, "out_logs_p"), (z_p, "out_z_p"), (z, "out_z")):
        assert metrics.max_abs(got, d[key]) <= LATENT_TOL, key
    assert metrics.max_abs(o, d["out_o"]) <= WAV_TOL
    assert metrics.log_mel_l1(o[:, 0], d["out_o"][:, 0]) <= MEL_TOL
    assert (f0.cpu() - d["out_f0"]).abs().max() <= 1e-3 * max(1.0, float(d["out_f0"].abs().max()))
    assert metrics.max_abs(energy, d["out_energy"]) <= 1e-3
    with pytest.raises(TypeError):
        net.infer(d["in_phonemes"], d["in_lengths"], noise=d["out_noise"], duration_control=d["out_duration"])
    with pytest.raises(RuntimeError):
        net.voice_conversion(torch.zeros(1, cfg.spec_channels, 8), torch.tensor([8]), torch.tensor([0]), torch.tensor([1]))


def test_frame_rate_convs_in_f16x2_match_the_goldens(gpu_model):
    """The frame-rate convs of the front end (frame-prior encoder, projection, the flow's WaveNets) carry an f16x2
    packing beside the bf16x3 one and take it for launches too large for the short-sequence kernel -- long batches, out
    of the CPU oracle's reach.  Forced on here (vits_set_front_end_f16x2(2)) for the small reference goldens: latents
    and waveform at the fp32 tolerances, durations and masks (decided at phoneme rate, bf16x3 always) bit-equal."""
    from vits_amd import _native as N
    N.lib.vits_set_front_end_f16x2(2)
    try:
        for name in ("ragged_b2", "predicted_scalar", "tensor_controls"):
            d, meta = load_case(name)
            st = {}
            o, x_mask, (z, z_p, m_p, logs_p), duration, *_ = gpu_model.infer(
                d["in_phonemes"], d["in_lengths"], sid=d["in_sid"], noise=d["out_noise"], stages=st, **infer_kwargs(d, meta))
            assert torch.equal(x_mask.cpu(), d["out_x_mask"])
            assert torch.equal(duration.cpu().to(torch.float64), d["out_duration"].to(torch.float64))
            for got, key in ((st["prior"], "stage_prior"), (m_p, "out_m_p"), (logs_p, "out_logs_p"), (z_p, "out_z_p"),
                             (z, "out_z")):
                assert metrics.max_abs(got, d[key]) <= LATENT_TOL, key
            assert metrics.max_abs(o, d["out_o"]) <= WAV_TOL
            assert metrics.log_mel_l1(o[:, 0], d["out_o"][:, 0]) <= MEL_TOL
        assert not gpu_model.range_overflow()
    finally:
        N.lib.vits_set_front_end_f16x2(1)


def test_front_end_beside_a_generator_picks_other_kernels_not_other_values(gpu_model):
    """vits_model_set_front_end_overlapped (what infer_batches sets for its front ends): at 16 x 3 s -- 4128 frames, beyond
    the 2048 from which the frame-rate convs then leave the short-sequence kernel for the f16x2 one -- the reference
    samples hold at the fp32 tolerance, durations and masks are the plain call's bytes, the waveform is within 1e-5 of
    the plain call's and NOT its bytes (the other kernels did run); throughput mode returns exactly these bytes."""
    from vits_amd import _native as N
    b = synthetic.config_c2()
    g = np.load(os.path.join(GOLDEN, "c2.npz"))
    kw = dict(sid=b.sid, noise=b.noise, noise_scale=b.noise_scale, duration_control=b.durations)
    plain = gpu_model.infer(b.phonemes, b.lengths, **kw)
    N.check(N.lib.vits_model_set_front_end_overlapped(gpu_model._handle, 1))
    try:
        hinted = gpu_model.infer(b.phonemes, b.lengths, **kw)
    finally:
        N.check(N.lib.vits_model_set_front_end_overlapped(gpu_model._handle, 0))
    assert np.abs(hinted[0][:, 0, ::97].cpu().numpy() - g["out_o_stride97"]).max() <= WAV_TOL
    assert torch.equal(hinted[1], plain[1]) and torch.equal(hinted[3], plain[3])
    assert metrics.max_abs(hinted[0], plain[0]) <= 1e-5 and not torch.equal(hinted[0], plain[0])
    piped = next(iter(gpu_model.infer_batches([dict(phonemes=b.phonemes, phonemes_lengths=b.lengths, **kw)])))
    assert torch.equal(piped[0], hinted[0])
    assert not gpu_model.range_overflow()


def test_infer_c4_corpus_slice(gpu_model, oracle_model):
    """BASELINE configs[3] (the 10 000-utterance corpus: lognormal lengths clipped to 88..1400 frames, sorted, cut
    into batches of <= 64): a fixed slice of its batches -- the longest, the shortest, the first ragged one and a
    ragged one from the middle -- through infer_batches, the path the corpus benchmark uses.
      * every batch: what plain infer() returns for it -- the same masks and frame counts, the waveform within
        1e-5 and the latent within 5e-5 (batches of 2048 frames and more run their frame-rate convs on the f16x2 packing in throughput mode, on
        the latency-oriented bf16x3 kernel in a plain call, where that one takes them; the same bytes otherwise);
      * the small ragged batch: the whole batch, padded tails included, against the CPU oracle (parity is defined per
        identical batch: padding leaks into valid positions, reference models.py:746-765);
      * the unpadded batches: a row equals its own batch-1 result, and that one against the oracle."""
    batches = synthetic.corpus_c4(10000)
    assert len(batches) == 157 and batches[0][1][0] == 88 and batches[-1][1][-1] == 1400
    ragged = next(i for i, (_, ts) in enumerate(batches) if len(set(ts)) > 1)
    picks = [155, 0, ragged, 78]                       # 155: the heaviest batch, 64 utterances of 1400 frames
    assert batches[155][1] == [1400] * 64
    assert len(set(batches[78][1])) > 1
    data = [synthetic.make_batch(1000 + i, *batches[i]) for i in picks]
    kws = [dict(phonemes=b.phonemes, phonemes_lengths=b.lengths, sid=b.sid, noise_scale=b.noise_scale,
                duration_control=b.durations, noise=b.noise) for b in data]
    got = list(gpu_model.infer_batches(kws))
    assert len(got) == len(picks)
    for b, kw, g in zip(data, kws, got):
        k = dict(kw)
        r = gpu_model.infer(k.pop("phonemes"), k.pop("phonemes_lengths"), **k)
        assert torch.equal(g[1], r[1])
        assert metrics.max_abs(g[0], r[0]) <= 1e-5 and metrics.max_abs(g[2][0], r[2][0]) <= 5e-5     # waveform, latent z
        assert g[0].shape == (len(b.frames), 1, int(b.frames.max()) * 512)
        assert torch.isfinite(g[0]).all() and float(g[0].abs().max()) <= 1.0
        assert torch.equal(g[1].sum(dim=(1, 2)).cpu(), b.frames)
        del r
    # the first ragged batch of the corpus, whole, against the oracle
    b, g = data[2], got[2]
    ref = oracle_model.infer(b.phonemes, b.lengths, sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations,
                             noise=b.noise)
    assert metrics.max_abs(g[0], ref[0]) <= WAV_TOL
    assert metrics.max_abs(g[2][0], ref[2][0]) <= LATENT_TOL
    for j in (0, len(b.frames) - 1):
        n = int(b.frames[j]) * 512
        assert metrics.log_mel_l1(g[0][j:j + 1, 0, :n], ref[0][j:j + 1, 0, :n]) <= MEL_TOL
    # unpadded batches (all rows of one length): row independence, then the oracle on that single utterance
    for bi, row in ((0, 17), (1, 40)):
        b, g = data[bi], got[bi]
        assert len(set(b.frames.tolist())) == 1
        kw1 = dict(sid=b.sid[row:row + 1], noise_scale=b.noise_scale, duration_control=b.durations[row:row + 1],
                   noise=b.noise[row:row + 1])
        o1 = gpu_model.infer(b.phonemes[row:row + 1], b.lengths[row:row + 1], **kw1)[0]
        assert metrics.max_abs(o1[0], g[0][row]) <= 1e-5
        ref1 = oracle_model.infer(b.phonemes[row:row + 1], b.lengths[row:row + 1], **kw1)[0]
        assert metrics.max_abs(o1, ref1) <= WAV_TOL
        assert metrics.log_mel_l1(o1[:, 0], ref1[:, 0]) <= MEL_TOL
    assert not gpu_model.range_overflow()


def test_corpus_batches_do_not_depend_on_the_rank_count(gpu_model, synthetic_sd):
    """The sharded corpus path (BASELINE configs[3] on N GPUs): batch composition is fixed before the batches are dealt
    to ranks, so every batch must come out with the same bytes whichever rank synthesises it, after whichever other
    batches.  One GPU plays the ranks in turn: N = 1 (one model, every batch in corpus order), then N = 2 and N = 4
    with a second model instance (its own workspaces, as another process would have) running each rank's share through
    infer_batches, the path the corpus benchmark uses."""
    from vits_amd import sharding
    batches = synthetic.corpus_c4(500)
    assert len(batches) >= 8 and any(len(set(ts)) > 1 for _, ts in batches)

    def kw_of(i):
        b = synthetic.make_batch(1000 + i, *batches[i])
        return dict(phonemes=b.phonemes, phonemes_lengths=b.lengths, sid=b.sid, noise_scale=b.noise_scale,
                    duration_control=b.durations, noise=b.noise)

    whole = [g[0].clone() for g in gpu_model.infer_batches([kw_of(i) for i in range(len(batches))])]
    other = _gpu_model(synthetic_sd)
    for world in (2, 4):
        seen = []
        for rank in range(world):
            mine = sharding.deal_batches(batches, world, rank)
            seen += mine
            for i, g in zip(mine, other.infer_batches([kw_of(i) for i in mine])):
                assert torch.equal(g[0], whole[i]), (world, rank, i)
        assert sorted(seen) == list(range(len(batches)))
    assert not gpu_model.range_overflow() and not other.range_overflow()


def test_generator_chain_with_unfusable_pair(synthetic_sd):
    """A ResBlock whose pairs do not all qualify for the fused-pair kernel (kernel 7 with dilation 9: the first conv's
    halo of 27 samples is beyond the fused kernel's 25) must still produce the reference's output: the chain is then
    run unfused as a whole.  Checked against the oracle with the same dilations."""
    import dataclasses
    from oracle.mvts_oracle import OracleSynthesizer
    dil = ((1, 3, 5), (1, 3, 9), (1, 2, 6))
    cfg = dataclasses.replace(synthetic.ModelConfig(), resblock_dilation_sizes=dil)
    net = _model(synthetic_sd, cfg=cfg)
    orc = OracleSynthesizer(synthetic_sd, resblock_dilation_sizes=dil)
    b = synthetic.make_batch(51, [11, 7], [37, 29])
    kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
    o = net.infer(b.phonemes, b.lengths, **kw)[0]
    ref = orc.infer(b.phonemes, b.lengths, **kw)[0]
    assert o.shape == ref.shape
    assert metrics.max_abs(o, ref) <= WAV_TOL
    assert metrics.log_mel_l1(o[:, 0], ref[:, 0]) <= MEL_TOL


def test_infer_resblock2_generator():
    """resblock "2" (modules.ResBlock2, reference modules.py:228-252: one conv per dilation, x = c(leaky_relu(x)) + x):
    against vectors of the reference built with that option, and streaming windows against the whole utterance."""
    from golden_util import infer_kwargs, load_case
    cfg = synthetic.ModelConfig(resblock="2", resblock_dilation_sizes=((1, 3), (1, 3), (1, 3)))
    sd = synthetic.synthetic_state_dict(cfg, include_enc_q=False)
    net = _model(sd, cfg=cfg)
    d, meta = load_case("resblock2_b2")
    kw = infer_kwargs(d, meta)
    o, x_mask, (z, *_), *_ = net.infer(d["in_phonemes"], d["in_lengths"], sid=d["in_sid"], noise=d["out_noise"], **kw)
    assert torch.equal(x_mask.cpu(), d["out_x_mask"])
    assert metrics.max_abs(z, d["out_z"]) <= LATENT_TOL
    assert o.shape == d["out_o"].shape and metrics.max_abs(o, d["out_o"]) <= WAV_TOL
    assert metrics.log_mel_l1(o[:, 0], d["out_o"][:, 0]) <= MEL_TOL
    st = net.infer_stream(d["in_phonemes"], d["in_lengths"], sid=d["in_sid"], noise=d["out_noise"], chunk_frames=16, **kw)
    assert st.halo_frames == 8
    assert metrics.max_abs(torch.cat([w for _, w in st], -1), o) <= STREAM_TOL


def test_infer_long_utterance_beyond_2048_frames(gpu_model, oracle_model):
    """An utterance of 2600 frames (30 s): past the 2048 positions the lanes-on-keys attention kernel can hold, the
    frame-rate encoder runs on the matrix-core attention kernel, which walks the keys in tiles and has no such bound.
    One row against the oracle, every returned tensor."""
    b = synthetic.make_batch(71, [130], [2600])
    kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
    got = gpu_model.infer(b.phonemes, b.lengths, **kw)
    ref = oracle_model.infer(b.phonemes, b.lengths, **kw)
    assert got[0].shape == ref[0].shape == (1, 1, 2600 * 512)
    assert metrics.max_abs(got[0], ref[0]) <= WAV_TOL
    assert metrics.log_mel_l1(got[0][:, 0], ref[0][:, 0]) <= MEL_TOL
    for name, x, y in zip(("z", "z_p", "m_p", "logs_p"), got[2], ref[2]):
        assert metrics.max_abs(x, y) <= LATENT_TOL, name


def test_layer_norm_in_the_convs_matches_separate_kernels(synthetic_sd, oracle_model):
    """The encoders' LayerNorms are not kernels by default: the conv that produces x + f(x) writes per-time-step
    statistics and the consumers normalise while they load.  Against the same model with every LayerNorm as its own
    launch (vits_model_set_ln_fused(0)): the same tensors to rounding, on a ragged batch (statistics of padded steps,
    masks) at a phoneme-rate and a frame-rate grid that take the split-K kernel and at a frame-rate grid large enough
    for the one-role kernel; and each against the oracle."""
    from vits_amd import _native as N
    net = _model(synthetic_sd)
    for seed, pc, fc in ((81, [23, 9, 17], [130, 41, 99]), (82, [60] * 40, [700] * 39 + [333])):
        b = synthetic.make_batch(seed, pc, fc)
        kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
        N.check(N.lib.vits_model_set_ln_fused(net._handle, 1), "set_ln_fused")
        fused = net.infer(b.phonemes, b.lengths, **kw)
        N.check(N.lib.vits_model_set_ln_fused(net._handle, 0), "set_ln_fused")
        plain = net.infer(b.phonemes, b.lengths, **kw)
        N.check(N.lib.vits_model_set_ln_fused(net._handle, 1), "set_ln_fused")
        for name, x, y in zip(("z", "z_p", "m_p", "logs_p"), fused[2], plain[2]):
            assert metrics.max_abs(x, y) <= 2e-5, name
        assert metrics.max_abs(fused[4], plain[4]) <= 1e-2                      # F0 in Hz
        assert metrics.max_abs(fused[0], plain[0]) <= 2e-5
        if len(pc) <= 4:
            ref = oracle_model.infer(b.phonemes, b.lengths, **kw)
            assert metrics.max_abs(fused[0], ref[0]) <= WAV_TOL
            for name, x, y in zip(("z", "z_p", "m_p", "logs_p"), fused[2], ref[2]):
                assert metrics.max_abs(x, y) <= LATENT_TOL, name


def test_f16x2_range_overflow_is_reported(synthetic_sd):
    """The default arithmetic (f16x2) clamps activations to the fp16 range.  A checkpoint whose generator leaves that
    range must be reported: range_overflow() turns true, the next infer warns once and names precision="bf16x3"
    (or raises with strict_range=True), and bf16x3 runs the same checkpoint without a report."""
    import warnings
    sd = dict(synthetic_sd)
    sd["dec.conv_pre.weight"] = sd["dec.conv_pre.weight"] * 1.0e5        # |weights| stay inside the fp16 range
    sd["dec.conv_pre.bias"] = sd["dec.conv_pre.bias"] * 1.0e5
    b = synthetic.make_batch(61, [8], [24])
    kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
    net = _model(sd)
    assert not net.range_overflow()
    net.infer(b.phonemes, b.lengths, **kw)
    assert net.range_overflow()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        net.infer(b.phonemes, b.lengths, **kw)
        net.infer(b.phonemes, b.lengths, **kw)
    msgs = [str(w.message) for w in rec if issubclass(w.category, RuntimeWarning)]
    assert len(msgs) == 1 and "bf16x3" in msgs[0] and "65504" in msgs[0]
    strict = _model(sd, strict_range=True)
    strict.infer(b.phonemes, b.lengths, **kw)
    with pytest.raises(OverflowError, match="bf16x3"):
        strict.infer(b.phonemes, b.lengths, **kw)
    wide = _model(sd, "bf16x3")
    wide.infer(b.phonemes, b.lengths, **kw)
    assert not wide.range_overflow()
    ok = _model(synthetic_sd)
    ok.infer(b.phonemes, b.lengths, **kw)
    assert not ok.range_overflow()


def test_model_is_exclusive_and_stream_safe(gpu_model):
    """One workspace serves every call: a second thread inside a call is refused, and a call issued on another
    stream than the previous one waits for it instead of overwriting its scratch."""
    b = synthetic.make_batch(71, [10, 6], [60, 33])
    kw = dict(sid=b.sid, noise_scale=b.noise_scale, duration_control=b.durations, noise=b.noise)
    ref = gpu_model.infer(b.phonemes, b.lengths, **kw)[0]
    assert gpu_model._busy.acquire(blocking=False)
    try:
        with pytest.raises(RuntimeError, match="re-entrant"):
            gpu_model.infer(b.phonemes, b.lengths, **kw)
        with pytest.raises(RuntimeError, match="re-entrant"):
            next(iter(gpu_model.infer_batches([dict(phonemes=b.phonemes, phonemes_lengths=b.lengths, **kw)])))
    finally:
        gpu_model._busy.release()
    side = torch.cuda.Stream()
    big = synthetic.make_batch(72, [50] * 8, [258] * 8)
    kwb = dict(sid=big.sid, noise_scale=big.noise_scale, duration_control=big.durations, noise=big.noise)
    with torch.cuda.stream(side):
        o_side = gpu_model.infer(big.phonemes, big.lengths, **kwb)[0]          # still running when the next call starts
    o_main = gpu_model.infer(b.phonemes, b.lengths, **kw)[0]
    torch.cuda.synchronize()
    assert torch.equal(o_main, ref)
    assert torch.equal(o_side, gpu_model.infer(big.phonemes, big.lengths, **kwb)[0])
