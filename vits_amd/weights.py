"""Load-time weight preparation: fold weight-norm, fuse / reorder, pack into MFMA fragment order, upload.

Input is a state dict in the reference checkpoint layout (reference utils.py:52-68: ``{'model': state_dict, ...}``
with old-style ``weight_g`` / ``weight_v`` pairs for weight-normed layers).  Keys the synthesis path never reads
(``enc_q.*``, ``enc_p.proj.*``, ``frame_prior_net.emb.weight``, ``energy_predictor.predictor.proj.*``) are ignored.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import _native as N

# enc_q.* is read by voice_conversion only: prepared when present, never required
UNUSED_PREFIXES = ("enc_q.", "enc_p.proj.", "frame_prior_net.emb.", "energy_predictor.predictor.proj.")


def fold_weight_norm(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """w = v * (g / ||v||) with the norm over all dims but 0 (torch.nn.utils.weight_norm default).  The reference
    re-evaluates this on every forward (it never calls remove_weight_norm); folding once is the same math."""
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / nrm)


def folded(sd: Dict[str, torch.Tensor], base: str) -> torch.Tensor:
    """The effective weight of layer `base`, whether stored plain or as a weight_g / weight_v pair."""
    if base + ".weight" in sd:
        return sd[base + ".weight"].float()
    if base + ".weight_v" in sd:
        return fold_weight_norm(sd[base + ".weight_v"].float(), sd[base + ".weight_g"].float())
    raise KeyError(f"{base}.weight (or weight_g / weight_v) is not in the checkpoint")


def pack_rows(w_rows: torch.Tensor, precision: int = N.PREC_F32) -> torch.Tensor:
    """[rows, cin, taps] fp32 (CPU) -> flat packed fragment order used by the conv kernel (fp32; three bf16
    planes stored as int16 bit patterns for the bf16x3 mode; one plane for the plain bf16 mode)."""
    w_rows = w_rows.contiguous().float()
    rows, cin, taps = w_rows.shape
    if precision == N.PREC_BF16X3:
        out = torch.empty(3 * rows * cin * taps, dtype=torch.int16)
        N.check(N.lib.vits_pack_conv_weight_bf16x3(w_rows.data_ptr(), rows, cin, taps, out.data_ptr()), "pack_bf16x3")
        return out
    if precision == N.PREC_F16X2:
        out = torch.empty(2 * rows * cin * taps, dtype=torch.int16)
        N.check(N.lib.vits_pack_conv_weight_f16x2(w_rows.data_ptr(), rows, cin, taps, out.data_ptr()), "pack_f16x2")
        return out
    if precision == N.PREC_BF16:
        out = torch.empty(rows * cin * taps, dtype=torch.int16)
        N.check(N.lib.vits_pack_conv_weight_bf16(w_rows.data_ptr(), rows, cin, taps, out.data_ptr()), "pack_bf16")
        return out
    out = torch.empty(rows * cin * taps, dtype=torch.float32)
    N.check(N.lib.vits_pack_conv_weight(w_rows.data_ptr(), rows, cin, taps, out.data_ptr()), "pack_conv_weight")
    return out


def transposed_conv_rows(wt: torch.Tensor, bias: torch.Tensor, stride: int, kernel: int, padding: int):
    """Polyphase rows of a ConvTranspose1d weight [Cin, Cout, k] (reference models.py:268-270).

    Output sample u*t + r of channel co is  sum_ci sum_j W[ci, co, ((r+p) mod u) + u*j] * x[ci, t + floor((r+p)/u) - j],
    so every (co, r) pair is one GEMM row with k/u taps.  Rows are ordered so that one 32-row MFMA tile holds rows
    of a single offset class floor((r+p)/u) and each lane ends up with consecutive output samples:
      stride 8: tile = 8 channels x phases {0..3} or {4..7};  stride 4: 8 channels x 4 phases;
      stride 2: 32 channels x one phase.
    Returns (w_rows [Cout*u, Cin, k/u], bias_rows [Cout*u], mode, taps)."""
    cin, cout, k = wt.shape
    u, p = stride, padding
    assert k == kernel and k % u == 0
    taps = k // u
    if (u, k, p) == (8, 16, 4):
        mode = N.MODE_UP8
        assert cout % 8 == 0
    elif (u, k, p) == (4, 4, 0):
        mode = N.MODE_UP4
        assert cout % 8 == 0
    elif (u, k, p) == (2, 4, 1):
        mode = N.MODE_UP2
        assert cout % 32 == 0
    else:
        raise NotImplementedError(f"ConvTranspose1d stride {u} kernel {k} padding {p}: supported are (8,16,4), (4,4,0), (2,4,1)")
    rows_co: List[int] = []
    rows_r: List[int] = []
    n_rows = cout * u
    for row in range(n_rows):
        mt, rit = divmod(row, 32)
        g, h, jj = rit // 8, (rit // 4) % 2, rit % 4
        if mode == N.MODE_UP8:
            cb, cls = divmod(mt, 2)
            co, r = cb * 8 + 2 * g + h, 4 * cls + jj
        elif mode == N.MODE_UP4:
            co, r = mt * 8 + 2 * g + h, jj
        else:
            cb, cls = divmod(mt, 2)
            co, r = cb * 32 + rit, cls
        rows_co.append(co)
        rows_r.append(r)
    co_t = torch.tensor(rows_co)
    r_t = torch.tensor(rows_r)
    w_rows = torch.empty(n_rows, cin, taps, dtype=torch.float32)
    for j in range(taps):
        kk = (r_t + p) % u + u * j                       # [rows]
        w_rows[:, :, j] = wt[:, co_t, kk].t()            # wt[:, co, k] -> [cin, rows] -> [rows, cin]
    return w_rows, bias[co_t].contiguous(), mode, taps


class PreparedWeights:
    """CPU-side result of preparing a checkpoint: plain tensors and packed convs keyed by library names."""

    def __init__(self, precision: int = N.PREC_F32):
        # model-level mode.  PREC_BF16 / PREC_F16X2 apply to the generator's convs ("dec.*", 97 % of the FLOPs); every
        # conv upstream of the latent z then runs in bf16x3 (full fp32 range and 24 bits where durations, masks and
        # latents are decided, and where the time is not)
        self.precision = precision
        self.plain: Dict[str, torch.Tensor] = {}
        self.convs: Dict[str, Tuple[torch.Tensor, torch.Tensor, int, int, int, int, int, int]] = {}
        # second packing (f16x2) of bf16x3 front-end convs that run at frame rate: the library takes it for launches
        # too large for its short-sequence kernel (vits_model_set_conv_alt)
        self.alts: Dict[str, torch.Tensor] = {}

    def conv_precision(self, name: str) -> int:
        if self.precision in (N.PREC_BF16, N.PREC_F16X2) and not name.startswith("dec."):
            return N.PREC_BF16X3
        return self.precision

    def add_plain(self, name: str, t: torch.Tensor):
        self.plain[name] = t.detach().float().contiguous().reshape(-1)

    def add_conv(self, name: str, w_rows: torch.Tensor, bias_rows, dilation: int = 1, mode: int = N.MODE_PLAIN):
        rows, cin, taps = w_rows.shape
        b = None if bias_rows is None else bias_rows.detach().float().contiguous()
        prec = self.conv_precision(name)
        self.convs[name] = (pack_rows(w_rows, prec), b, cin, rows, taps, dilation, mode, prec)
        if (prec == N.PREC_BF16X3 and self.precision in (N.PREC_BF16, N.PREC_F16X2) and name.startswith(FRAME_RATE_CONVS)
                and mode in (N.MODE_PLAIN, N.MODE_GATE) and rows % 64 == 0 and cin % 16 == 0):
            try:
                self.alts[name] = pack_rows(w_rows, N.PREC_F16X2)
            except Exception:                    # a weight outside the fp16 range: this conv stays bf16x3 at every size
                pass


# convs that run once per FRAME (frame-prior encoder, projection, coupling flow): at long batches they are matrix-bound
# and get an f16x2 packing beside the bf16x3 one; everything at phoneme rate (text encoder, the predictors: where
# durations and masks are decided) has bf16x3 only
FRAME_RATE_CONVS = ("frame_prior_net.fft_block.", "project", "flow.")

ENCODERS = (("enc_p.encoder", "n_layers"), ("pitch_predictor.pitch_net", "pitch_layers"),
            ("frame_prior_net.fft_block", "n_layers"))


def prepare(sd: Dict[str, torch.Tensor], cfg, precision: int = N.PREC_F32) -> PreparedWeights:
    """cfg: vits_amd.synthetic.ModelConfig-like object with the model dimensions."""
    sd = {k: v.detach().cpu() for k, v in sd.items()}
    pw = PreparedWeights(precision)
    C, G = cfg.hidden_channels, cfg.gin_channels

    def bias(base):
        return sd[base + ".bias"].float()

    pw.add_plain("enc_p.symbol_emb", sd["enc_p.symbol_emb.weight"])
    # a speaker-less model (n_speakers == 0; reference models.py:719-722: g = None) has no speaker table and applies
    # none of its conditioning convs, whether the checkpoint holds them (gin_channels > 0) or not
    speakers = cfg.n_speakers > 0
    if speakers:
        pw.add_plain("emb_g", sd["emb_g.weight"])
    # attentions.Encoder stacks (reference attentions.py:10-44)
    for prefix, nl_attr in ENCODERS:
        for i in range(getattr(cfg, nl_attr)):
            a, f, o = f"{prefix}.attn_layers.{i}", f"{prefix}.ffn_layers.{i}", f"{prefix}.{i}"
            wq = torch.cat([folded(sd, f"{a}.conv_{n}") for n in "qkv"], 0)
            bq = torch.cat([bias(f"{a}.conv_{n}") for n in "qkv"], 0)
            pw.add_conv(o + ".qkv", wq, bq)
            pw.add_conv(o + ".o", folded(sd, a + ".conv_o"), bias(a + ".conv_o"))
            pw.add_plain(o + ".emb_rel_k", sd[a + ".emb_rel_k"][0])
            pw.add_plain(o + ".emb_rel_v", sd[a + ".emb_rel_v"][0])
            pw.add_conv(o + ".ffn1", folded(sd, f + ".conv_1"), bias(f + ".conv_1"))
            pw.add_conv(o + ".ffn2", folded(sd, f + ".conv_2"), bias(f + ".conv_2"))
            for n in (1, 2):
                pw.add_plain(f"{o}.ln{n}.g", sd[f"{prefix}.norm_layers_{n}.{i}.gamma"])
                pw.add_plain(f"{o}.ln{n}.b", sd[f"{prefix}.norm_layers_{n}.{i}.beta"])
    # duration predictor (reference models.py:105-139)
    d = "duration_predictor"
    pw.add_conv("dp.conv1", folded(sd, d + ".conv_1"), bias(d + ".conv_1"))
    pw.add_conv("dp.conv2", folded(sd, d + ".conv_2"), bias(d + ".conv_2"))
    for n in (1, 2):
        pw.add_plain(f"dp.ln{n}.g", sd[f"{d}.norm_{n}.gamma"])
        pw.add_plain(f"dp.ln{n}.b", sd[f"{d}.norm_{n}.beta"])
    pw.add_plain("dp.proj.w", sd[d + ".proj.weight"])
    pw.add_plain("dp.proj.b", sd[d + ".proj.bias"])
    # pitch / energy heads and prenets (reference models.py:543, 657-658; frame_prior_network.py:58-109)
    pw.add_plain("pitch.proj.w", sd["pitch_predictor.proj_f0.weight"])
    pw.add_plain("pitch.proj.b", sd["pitch_predictor.proj_f0.bias"])
    pw.add_plain("pitch_prenet.w", sd["pitch_prenet.weight"])
    pw.add_plain("pitch_prenet.b", sd["pitch_prenet.bias"])
    e = "energy_predictor.predictor"
    pw.add_conv("energy.conv1", folded(sd, e + ".conv_layer.conv_1.conv"), bias(e + ".conv_layer.conv_1.conv"))
    pw.add_conv("energy.conv2", folded(sd, e + ".conv_layer.conv_2.conv"), bias(e + ".conv_layer.conv_2.conv"))
    for n in (1, 2):
        pw.add_plain(f"energy.ln{n}.g", sd[f"{e}.conv_layer.layer_norm_{n}.weight"])
        pw.add_plain(f"energy.ln{n}.b", sd[f"{e}.conv_layer.layer_norm_{n}.bias"])
    pw.add_plain("energy.lin.w", sd[e + ".linear_layer.weight"])
    pw.add_plain("energy.lin.b", sd[e + ".linear_layer.bias"])
    pw.add_plain("energy_prenet.w", sd["energy_prenet.weight"])
    pw.add_plain("energy_prenet.b", sd["energy_prenet.bias"])
    pw.add_conv("project", folded(sd, "project.proj"), bias("project.proj"))
    # conditioning: every 1x1 conv applied to g, stacked into one matrix (order fixed by the library)
    cw, cb = [], []
    if speakers:
        cw = [folded(sd, "duration_predictor.cond")[:, :, 0], folded(sd, "pitch_predictor.cond")[:, :, 0],
              folded(sd, "energy_predictor.cond")[:, :, 0], folded(sd, "dec.cond")[:, :, 0]]
        cb = [bias("duration_predictor.cond"), bias("pitch_predictor.cond"), bias("energy_predictor.cond"),
              bias("dec.cond")]
    # reverse coupling flow (reference models.py:184-216, modules.py:107-172, 294-339)
    nf, nl = cfg.n_flows, cfg.flow_layers
    for f in range(nf):
        p, o = f"flow.flows.{2 * f}", f"flow.{f}"
        flipped = (nf - f) % 2 == 1        # number of channel flips seen before this layer in the reverse pass
        w_pre, b_pre = folded(sd, p + ".pre"), bias(p + ".pre")
        w_post, b_post = folded(sd, p + ".post"), bias(p + ".post")
        if flipped:                        # fold torch.flip(x, [1]) into the 1x1 convs at both ends
            w_pre = torch.flip(w_pre, [1])
            w_post, b_post = torch.flip(w_post, [0]), torch.flip(b_post, [0])
        pw.add_conv(o + ".pre", w_pre, b_pre)
        pw.add_conv(o + ".post", w_post, b_post)
        for l in range(nl):
            base = f"{p}.enc.in_layers.{l}"
            w_in = folded(sd, base)
            pw.add_conv(f"{o}.in.{l}", w_in, bias(base), mode=N.MODE_GATE)
            base = f"{p}.enc.res_skip_layers.{l}"
            w_rs, b_rs = folded(sd, base), bias(base)
            if l + 1 < nl:
                pw.add_conv(f"{o}.res.{l}", w_rs[:C], b_rs[:C])
                pw.add_conv(f"{o}.skip.{l}", w_rs[C:], b_rs[C:])
                pw.add_conv(f"{o}.rs.{l}", w_rs, b_rs)        # both halves as the reference's single conv: one launch
            else:
                pw.add_conv(f"{o}.skip.{l}", w_rs, b_rs)
        if speakers:
            cw.append(folded(sd, p + ".enc.cond_layer")[:, :, 0])
            cb.append(bias(p + ".enc.cond_layer"))
    # posterior encoder (reference models.py:221-250; voice conversion only): same WN as a flow layer with
    # cfg.enc_q_layers layers; the spectrogram channels of `pre` are zero-padded to a multiple of 16
    if "enc_q.pre.weight" in sd:
        w_pre = folded(sd, "enc_q.pre")
        pad = -w_pre.shape[1] % 16
        pw.add_conv("enc_q.pre", torch.nn.functional.pad(w_pre, (0, 0, 0, pad)), bias("enc_q.pre"))
        pw.add_conv("enc_q.proj", folded(sd, "enc_q.proj"), bias("enc_q.proj"))
        nlq = cfg.enc_q_layers
        for l in range(nlq):
            base = f"enc_q.enc.in_layers.{l}"
            pw.add_conv(f"enc_q.in.{l}", folded(sd, base), bias(base), mode=N.MODE_GATE)
            base = f"enc_q.enc.res_skip_layers.{l}"
            w_rs, b_rs = folded(sd, base), bias(base)
            if l + 1 < nlq:
                pw.add_conv(f"enc_q.res.{l}", w_rs[:C], b_rs[:C])
                pw.add_conv(f"enc_q.skip.{l}", w_rs[C:], b_rs[C:])
                pw.add_conv(f"enc_q.rs.{l}", w_rs, b_rs)
            else:
                pw.add_conv(f"enc_q.skip.{l}", w_rs, b_rs)
        if speakers:
            cw.append(folded(sd, "enc_q.enc.cond_layer")[:, :, 0])
            cb.append(bias("enc_q.enc.cond_layer"))
    if speakers:
        pw.add_plain("cond.w", torch.cat(cw, 0))
        pw.add_plain("cond.b", torch.cat(cb, 0))
    # generator (reference models.py:257-303, modules.py:183-225)
    pw.add_conv("dec.conv_pre", folded(sd, "dec.conv_pre"), bias("dec.conv_pre"))
    n_k = len(cfg.resblock_kernel_sizes)
    for i, (u, ku) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        w_rows, b_rows, mode, _ = transposed_conv_rows(folded(sd, f"dec.ups.{i}"), bias(f"dec.ups.{i}"), u, ku,
                                                       (ku - u) // 2)
        pw.add_conv(f"dec.ups.{i}", w_rows, b_rows, mode=mode)
        for j, dils in enumerate(cfg.resblock_dilation_sizes):
            r = i * n_k + j
            for q, dil in enumerate(dils):
                if str(cfg.resblock) == "2":                    # modules.ResBlock2: one conv per dilation
                    pw.add_conv(f"dec.rb.{r}.c1.{q}", folded(sd, f"dec.resblocks.{r}.convs.{q}"),
                                bias(f"dec.resblocks.{r}.convs.{q}"), dilation=dil)
                    continue
                pw.add_conv(f"dec.rb.{r}.c1.{q}", folded(sd, f"dec.resblocks.{r}.convs1.{q}"),
                            bias(f"dec.resblocks.{r}.convs1.{q}"), dilation=dil)
                pw.add_conv(f"dec.rb.{r}.c2.{q}", folded(sd, f"dec.resblocks.{r}.convs2.{q}"),
                            bias(f"dec.resblocks.{r}.convs2.{q}"), dilation=1)
    pw.add_plain("dec.conv_post.w", sd["dec.conv_post.weight"])
    return pw
