"""CPU-side checks: the shared library loads and exports the whole C ABI, weight packing / polyphase row order is
right (emulated in numpy against torch's conv ops), and the config / checkpoint helpers behave like the
reference's (reference utils.py:19-49, 217-223, 280-309)."""
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vits_amd import _native as N
from vits_amd import synthetic, utils, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vits_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(vits_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(N.lib, name), f"{name} declared in include/vits_amd.h but not exported"
    assert declared == set(N.EXPORTED)
    assert b"gfx950" in N.lib.vits_version()


def unpack(packed, rows, cin, taps):
    """Inverse of the fragment order documented in csrc/common.h: [cin/16][taps][2][rows/32][64 lanes][4]."""
    p = packed.numpy().reshape(cin // 16, taps, 2, rows // 32, 64, 4)
    w = np.zeros((rows, cin, taps), dtype=np.float32)
    for lane in range(64):
        for e in range(4):
            r, par = lane & 31, lane >> 5
            # row = mt*32 + r ; ci = chunk*16 + sq*8 + e*2 + par
            blk = p[:, :, :, :, lane, e]                     # [chunk, tap, sq, mt]
            for sq in range(2):
                ci = np.arange(cin // 16) * 16 + sq * 8 + e * 2 + par
                w[r::32, :, :][:, ci, :] = np.transpose(blk[:, :, sq, :], (2, 0, 1))
    return w


def test_pack_roundtrip():
    torch.manual_seed(0)
    w = torch.randn(64, 32, 5)
    assert np.array_equal(unpack(weights.pack_rows(w), 64, 32, 5), w.numpy())


def test_pack_bf16_planes():
    """bf16 fragment order [cin/16][taps][rows/32][plane][64 lanes][8]: lane l of m-tile mt holds
    W[mt*32 + (l & 31)][16*chunk + 8*(l >> 5) + j][tap]; the plain mode stores the round-to-nearest-even bf16 value,
    the bf16x3 mode three planes that sum back to the fp32 value exactly."""
    torch.manual_seed(2)
    rows, cin, taps = 64, 32, 3
    w = torch.randn(rows, cin, taps)
    w[0, 0, 0] = 1.0 + 2.0 ** -8             # exactly half way between two bf16 values: ties go to even (1.0)
    w[1, 0, 0] = 1.0 + 3 * 2.0 ** -8         # tie, rounds up to even (1 + 2^-6)

    def values(packed, planes):
        bits = packed.view(torch.int16).to(torch.int32) & 0xffff
        f = (bits << 16).view(torch.float32) if bits.dtype == torch.int32 else None
        f = f.reshape(cin // 16, taps, rows // 32, planes, 64, 8).double()
        out = torch.zeros(planes, rows, cin, taps, dtype=torch.float64)
        for lane in range(64):
            for j in range(8):
                ci = torch.arange(cin // 16) * 16 + 8 * (lane >> 5) + j
                # f[:, :, :, p, lane, j] is [chunk, tap, mt]
                for p_ in range(planes):
                    out[p_, (lane & 31)::32][:, ci, :] = f[:, :, :, p_, lane, j].permute(2, 0, 1)
        return out

    one = values(weights.pack_rows(w, N.PREC_BF16), 1)[0]
    assert torch.equal(one.float(), w.to(torch.bfloat16).float())
    assert one[0, 0, 0] == 1.0 and one[1, 0, 0] == 1.0 + 2.0 ** -6
    three = values(weights.pack_rows(w, N.PREC_BF16X3), 3)
    assert torch.equal(three.sum(0).float(), w)
    assert torch.equal(three[0].float(), one.float())


def test_pack_f16x2_planes():
    """fp16 fragment order [cin/16][taps][rows/32][2 planes][64 lanes][8]: h = fp16(w), l = fp16(w - h), both round to
    nearest even with subnormals kept, so h + l reproduces w to 2^-22 relative (2^-25 absolute below 1/2); a weight
    outside the fp16 range is refused."""
    torch.manual_seed(3)
    rows, cin, taps = 32, 16, 2
    w = torch.randn(rows, cin, taps) * 0.05
    w[0, 0, 0] = 3.0e-6                     # h is an fp16 subnormal
    w[1, 0, 0] = 1000.123
    packed = weights.pack_rows(w, N.PREC_F16X2)
    f = packed.view(torch.float16).reshape(cin // 16, taps, rows // 32, 2, 64, 8).double()
    got = torch.zeros(2, rows, cin, taps, dtype=torch.float64)
    for lane in range(64):
        for j in range(8):
            for p_ in range(2):
                got[p_, lane & 31, 8 * (lane >> 5) + j, :] = f[0, :, 0, p_, lane, j]
    h, l = got
    assert torch.equal(h.float(), w.to(torch.float16).float())
    assert torch.equal(l.float(), (w - w.to(torch.float16).float()).to(torch.float16).float())
    err = (h + l - w.double()).abs()
    assert float((err / w.double().abs().clamp_min(0.5)).max()) <= 2.0 ** -22
    assert float(err[w.abs() < 0.5].max()) <= 2.0 ** -25
    bad = w.clone()
    bad[2, 3, 1] = 7.0e4
    with pytest.raises(RuntimeError):
        weights.pack_rows(bad, N.PREC_F16X2)


def emulate_transposed(w_rows, bias_rows, mode, x, up):
    """What the conv kernel computes for a transposed conv: GEMM rows with per-class tap offsets, then the
    row -> (channel, phase) scatter of its epilogue."""
    rows, cin, taps = w_rows.shape
    cout, L = rows // up, x.shape[-1]
    xp = F.pad(x, (2, 2)).numpy()
    out = np.zeros((x.shape[0], cout, L * up), dtype=np.float64)
    for row in range(rows):
        mt, rit = divmod(row, 32)
        g, h, jj = rit // 8, (rit // 4) % 2, rit % 4
        if mode == N.MODE_UP8:
            cb, cls = divmod(mt, 2); co, r = cb * 8 + 2 * g + h, 4 * cls + jj
        elif mode == N.MODE_UP4:
            cls = 0; co, r = mt * 8 + 2 * g + h, jj
        else:
            cb, cls = divmod(mt, 2); co, r = cb * 32 + rit, cls
        acc = np.zeros((x.shape[0], L))
        for j in range(taps):
            off = (cls - j) if mode != N.MODE_UP4 else 0       # toff0 = class, tstep = -1
            acc += np.einsum("c,bct->bt", w_rows[row, :, j].numpy().astype(np.float64), xp[:, :, 2 + off:2 + off + L])
        out[:, co, r::up] = acc + float(bias_rows[row])
    return out


@pytest.mark.parametrize("u,k,cin,cout", [(8, 16, 16, 16), (4, 4, 16, 16), (2, 4, 16, 32)])
def test_polyphase_rows_match_conv_transpose1d(u, k, cin, cout):
    torch.manual_seed(1)
    wt, b, x = torch.randn(cin, cout, k), torch.randn(cout), torch.randn(2, cin, 9)
    w_rows, b_rows, mode, taps = weights.transposed_conv_rows(wt, b, u, k, (k - u) // 2)
    ref = F.conv_transpose1d(x.double(), wt.double(), b.double(), stride=u, padding=(k - u) // 2).numpy()
    got = emulate_transposed(w_rows, b_rows, mode, x, u)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-5


def test_prepare_covers_every_used_key(synthetic_sd):
    cfg = synthetic.ModelConfig()
    pw = weights.prepare(synthetic_sd, cfg)
    base = 14 * 4 + 2 + 2 + 1 + 4 * (2 + 4 + 3 + 4 + 3) + 1 + 4 + 72     # per flow: pre, post, 4 in, 3 res, 4 skip, 3 res+skip
    no_q = {k: v for k, v in synthetic_sd.items() if not k.startswith("enc_q.")}
    assert len(weights.prepare(no_q, cfg).convs) == base
    # with the posterior encoder's weights: pre, proj, 16 gated convs, 15 res + 16 skip 1x1 convs, 6144 cond rows
    assert "enc_q.pre.weight" in synthetic_sd and len(pw.convs) == base + 2 + 16 + 15 + 16 + 15
    assert pw.convs["enc_q.pre"][2:5] == (1040, 192, 1)
    flow0 = pw.convs["flow.0.in.0"]
    assert flow0[2:] == (192, 384, 5, 1, N.MODE_GATE, N.PREC_F32)
    # model-level "bf16": only the generator's convs are plain bf16, everything upstream of z stays fp32-grade
    pb = weights.prepare(synthetic_sd, cfg, N.PREC_BF16)
    assert {v[7] for k, v in pb.convs.items() if k.startswith("dec.")} == {N.PREC_BF16}
    assert {v[7] for k, v in pb.convs.items() if not k.startswith("dec.")} == {N.PREC_BF16X3}
    ph = weights.prepare(synthetic_sd, cfg, N.PREC_F16X2)
    assert {v[7] for k, v in ph.convs.items() if k.startswith("dec.")} == {N.PREC_F16X2}
    assert {v[7] for k, v in ph.convs.items() if not k.startswith("dec.")} == {N.PREC_BF16X3}
    assert pw.plain["cond.w"].numel() == (3 * 192 + 512 + 4 * 1536 + 16 * 384) * 256


def test_hparams_and_config_reader(tmp_path):
    text = '{\n "train": {"seed": 1234}, // trailing comment\n /* block\n comment */ "model": {"n_heads": 2, "url": "http://x/y"}\n}'
    p = tmp_path / "config.json"
    p.write_bytes(text.encode("gbk"))
    hps = utils.get_hparams_from_file(str(p))
    assert hps.train.seed == 1234 and hps.model["n_heads"] == 2 and "model" in hps and len(hps) == 2
    assert hps.model.url == "http://x/y"
    p.write_text(json.dumps({"a": {"b": 1}}))
    assert utils.get_hparams_from_file(str(p)).a.b == 1


def test_generator_halo_frames():
    """Receptive field of the generator in frames per side (what a streaming window needs as context): 13 for the
    reference configuration (reference configs/config.json:140-160), wider kernels / dilations widen it."""
    from vits_amd.models import generator_halo_frames
    cfg = synthetic.ModelConfig()
    assert generator_halo_frames(cfg) == 13
    import dataclasses
    wide = dataclasses.replace(cfg, resblock_kernel_sizes=(3, 7, 13))
    assert generator_halo_frames(wide) > 13
    # modules.ResBlock2: one conv per dilation, no second undilated conv -- 20 samples per stage for kernel 11 with
    # dilations (1, 3): 23 -> 13 -> 33 -> 10 -> 30 -> 5 -> 25 -> 5 -> 8 frames
    single = dataclasses.replace(cfg, resblock="2", resblock_dilation_sizes=((1, 3),) * 3)
    assert generator_halo_frames(single) == 8


def _cpu_model(monkeypatch):
    """A SynthesizerTrn whose upload step (the only part that needs a GPU) is recorded instead of run."""
    from vits_amd import SynthesizerTrn
    cfg = synthetic.ModelConfig()
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         **cfg.kwargs()).eval()
    uploads = []
    monkeypatch.setattr(net, "_upload", lambda pw=None: uploads.append(len(net._sd)))
    return net, uploads


def test_load_checkpoint_end_to_end(tmp_path, monkeypatch, synthetic_sd, capsys):
    """utils.load_checkpoint on a file written the way the reference writes it (reference utils.py:52-68:
    {'model', 'iteration', 'optimizer', 'learning_rate'}) and its tolerant key-by-key behaviour (utils.py:19-49):
    keys the file lacks or holds with another shape are reported and left to the model, which accepts that for
    tensors infer never reads and refuses it for tensors it needs."""
    net, uploads = _cpu_model(monkeypatch)
    path = str(tmp_path / "G_1000.pth")
    torch.save({"model": dict(synthetic_sd), "iteration": 1000, "optimizer": {"state": {}, "param_groups": []},
                "learning_rate": 2e-4}, path)
    model, opt, lr, it = utils.load_checkpoint(path, net, None)
    assert model is net and opt is None and lr == 2e-4 and it == 1000
    assert uploads == [len(synthetic_sd)] and set(net._sd) == set(synthetic_sd)
    assert torch.equal(net._sd["dec.conv_post.weight"], synthetic_sd["dec.conv_post.weight"])
    assert "is not in the checkpoint" not in capsys.readouterr().out

    # optional keys absent (enc_q.*, enc_p.proj.*: present in reference checkpoints, never read by infer)
    optional = {k for k in synthetic_sd if k.startswith(weights.UNUSED_PREFIXES)}
    assert optional and any(k.startswith("enc_q.") for k in optional)
    torch.save({"model": {k: v for k, v in synthetic_sd.items() if k not in optional}, "iteration": 7,
                "learning_rate": 1e-4}, path)
    fresh, _ = _cpu_model(monkeypatch)
    _, _, lr, it = utils.load_checkpoint(path, fresh, None)
    assert (lr, it) == (1e-4, 7) and set(fresh._sd) == set(synthetic_sd) - optional
    # a model that already holds a value keeps it, as the reference's module keeps its current parameter
    utils.load_checkpoint(path, net, None)
    assert set(net._sd) == set(synthetic_sd)
    out = capsys.readouterr().out
    assert f"error, {sorted(optional)[0]} is not in the checkpoint" in out

    # a required key absent: reported like the reference does, then refused (an uninitialised weight is not audio)
    torch.save({"model": {k: v for k, v in synthetic_sd.items() if k != "dec.conv_pre.weight"}, "iteration": 1,
                "learning_rate": 0.0}, path)
    with pytest.raises(RuntimeError, match="dec.conv_pre.weight"):
        utils.load_checkpoint(path, _cpu_model(monkeypatch)[0], None)
    assert "error, dec.conv_pre.weight is not in the checkpoint" in capsys.readouterr().out

    # a mis-shaped key is treated like an absent one (the reference's bare except, utils.py:38-41)
    bad = dict(synthetic_sd)
    bad["emb_g.weight"] = torch.zeros(3, 5)
    torch.save({"model": bad, "iteration": 1, "learning_rate": 0.0}, path)
    with pytest.raises(RuntimeError, match="emb_g.weight"):
        utils.load_checkpoint(path, _cpu_model(monkeypatch)[0], None)
    assert "error, emb_g.weight is not in the checkpoint" in capsys.readouterr().out
    bad = dict(synthetic_sd)
    bad["enc_p.proj.weight"] = torch.zeros(3, 5)          # optional and mis-shaped: reported, load goes through
    torch.save({"model": bad, "iteration": 2, "learning_rate": 0.0}, path)
    fresh, _ = _cpu_model(monkeypatch)
    assert utils.load_checkpoint(path, fresh, None)[3] == 2
    assert "enc_p.proj.weight" not in fresh._sd

    # a bare state dict (no 'model' wrapper) loads too
    torch.save(dict(synthetic_sd), path)
    assert utils.load_checkpoint(path, net, None)[2:] == (0.0, 0)


def test_model_handles_are_checked():
    """The C ABI hands out raw pointers: destroying one twice, destroying a pointer the library never made, or
    calling an entry point with either must be an error, not a double free."""
    import ctypes as C
    cfg = synthetic.ModelConfig()
    from vits_amd import SynthesizerTrn
    net = SynthesizerTrn(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                         **cfg.kwargs())
    h = C.c_void_p(net._handle.value)
    assert N.lib.vits_model_alive(h) == 1
    N.lib.vits_model_destroy(h)
    assert N.lib.vits_model_alive(h) == 0
    N.lib.vits_model_destroy(h)                                   # second destroy of the same pointer: ignored
    assert b"not a live model handle" in N.lib.vits_last_error()
    assert N.lib.vits_model_finalize(h) == -1 and N.lib.vits_cond_rows(h) == -1
    assert N.lib.vits_encoder_workspace_bytes(h, 1, 8) == -1
    junk = C.create_string_buffer(4096)                           # a pointer the library never created
    N.lib.vits_model_destroy(C.cast(junk, C.c_void_p))
    assert N.lib.vits_model_alive(C.cast(junk, C.c_void_p)) == 0
    del net                                                       # __del__ destroys again: must be harmless


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` started plainly launches N children with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set (the reference's train.py:44-48 spawns its workers itself, too), and a rank refuses to report a
    GPU count other than --gpus."""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            started.append((cmd, env, stdout))

        def poll(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)

    # the parent starts its ranks from its own process image, so it must not have touched the GPU runtime: every
    # torch.cuda entry that could initialise it is made to fail while spawn_ranks runs
    def touched(*a, **k):
        raise AssertionError("spawn_ranks called into torch.cuda")
    for name in ("device_count", "is_available", "init", "set_device", "current_device", "synchronize",
                 "get_device_properties", "mem_get_info"):
        monkeypatch.setattr(bench.torch.cuda, name, touched)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--backend", "gloo"])
    assert bench.spawn_ranks(argparse.Namespace(gpus=4, backend="gloo")) == 0
    assert len(started) == 4
    ports = {env["MASTER_PORT"] for _, env, _ in started}
    assert len(ports) == 1
    for r, (cmd, env, stdout) in enumerate(started):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--backend", "gloo"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "4", "127.0.0.1")
        assert (stdout is None) == (r == 0)                       # rank 0 prints the JSON line
    started.clear()
    assert bench.spawn_ranks(argparse.Namespace(gpus=2, backend="nccl")) == 0 and len(started) == 2
    monkeypatch.undo()
    # a rank whose WORLD_SIZE disagrees with --gpus stops instead of printing a line with the wrong n_gpus
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit, match="WORLD_SIZE"):
        bench.main()
    # a rank with no GPU under its LOCAL_RANK says so itself (RCCL needs one GPU per rank)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="needs a GPU per rank"):
        bench.main()


def test_bench_sets_up_the_process_group_after_the_models_streams(monkeypatch):
    """A rank of bench.py builds the model and puts work on its streams before it initialises the process group: a
    communicator's streams that came first would take hardware queues the model's four streams need (the step was a
    quarter slower that way).  And the model side notices a live RCCL group only when there is one."""
    import importlib.util
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("bench_order_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    order = []

    class Stop(Exception):
        pass

    def init(*a, **k):
        order.append("init_process_group")
        raise Stop()
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(bench.torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(bench, "build_model", lambda device, precision=None: order.append("build_model") or object())
    monkeypatch.setattr(bench, "claim_streams", lambda net, device: order.append("claim_streams"))
    monkeypatch.setattr(dist, "init_process_group", init)
    for k, v in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1"), ("VITS_BENCH_FORCE_DIST", "1")):
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("VITS_BENCH_DIST_EARLY", raising=False)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "1"])
    with pytest.raises(Stop):
        bench.main()
    assert order == ["build_model", "claim_streams", "init_process_group"]

    from vits_amd import models
    assert models._rccl_group_live() is False
