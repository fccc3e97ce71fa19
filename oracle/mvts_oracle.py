"""CPU oracle for the MVTS synthesis path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch fp32 restatement of the reference's ``SynthesizerTrn.infer`` (reference models.py:717-767),
written from the math of each stage with the file:line it follows cited per function.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module, and only as the
checker / the timed CPU baseline; nothing under ``vits_amd/`` imports it.

Parity pinning: the reference ships no tests or golden vectors for this path.  This restatement is pinned by
outputs of the reference itself, run in the build container by ``tools/make_goldens.py`` (which imports the
reference from /root/reference) on deterministic synthetic weights; the resulting vectors are committed under
``tests/golden/`` and ``tests/test_oracle_golden.py`` checks this file against them.

Differences from the reference that do not change results beyond fp32 rounding:
  * weight-norm (g * v / ||v||) is folded once at construction instead of on every call;
  * relative-position attention is evaluated in its banded form (|j - i| <= 4) instead of through
    zero-padded [L, 2L-1] matmuls (reference attentions.py:259-321);
  * the length regulator is a repeat_interleave instead of a Python double loop.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1          # reference modules.py:12


def fold_weight_norm(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """w = g * v / ||v||, norm over every dim but 0 (torch.nn.utils.weight_norm, dim=0).

    For ConvTranspose1d the weight is [Cin, Cout, k], so the norm is per INPUT channel
    (reference models.py:268-270)."""
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / nrm)


def sequence_mask(lengths: torch.Tensor, max_len: int) -> torch.Tensor:
    """reference commons.py:119-123"""
    return torch.arange(max_len, dtype=lengths.dtype)[None, :] < lengths[:, None]


def channel_layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """LayerNorm over the channel dim of [B, C, L] (reference modules.py:15-27)."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), gamma, beta, eps).transpose(1, 2)


class OracleSynthesizer:
    def __init__(self, state_dict: Dict[str, torch.Tensor], n_heads: int = 2, window: int = 4,
                 upsample_rates=(8, 8, 4, 2), upsample_kernel_sizes=(16, 16, 4, 4),
                 resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5),) * 3,
                 kernel_size: int = 3, n_speakers: int = 200, generator_bf16: bool = False, resblock: str = "1"):
        # generator_bf16: emulate the build's "bf16" conv mode -- both operands of every MFMA conv of the generator
        # (conv_pre, ups, ResBlock convs; not conv_post) rounded to nearest-even bf16, products accumulated in fp32
        self.generator_bf16 = generator_bf16
        self.resblock = str(resblock)             # "1": modules.ResBlock1, "2": modules.ResBlock2
        sd = {k: v.detach().to(torch.float32) for k, v in state_dict.items()}
        w: Dict[str, torch.Tensor] = {}
        for k, v in sd.items():
            if k.endswith(".weight_v"):
                w[k[:-2]] = fold_weight_norm(v, sd[k[:-1] + "g"])
            elif not k.endswith(".weight_g"):
                w[k] = v
        self.w = w
        self.n_heads = n_heads
        self.window = window
        self.ups = list(zip(upsample_rates, upsample_kernel_sizes))
        self.rb_k = list(resblock_kernel_sizes)
        self.rb_d = [list(d) for d in resblock_dilation_sizes]
        self.kernel_size = kernel_size
        self.n_speakers = n_speakers
        self.n_enc = {p: 1 + max(int(k.split(".attn_layers.")[1].split(".")[0]) for k in w if k.startswith(p + ".attn_layers."))
                      for p in ("enc_p.encoder", "pitch_predictor.pitch_net", "frame_prior_net.fft_block")}
        self.flow_ids = sorted({int(k.split(".")[2]) for k in w if k.startswith("flow.flows.")})
        self.wn_layers = 1 + max(int(k.split(".in_layers.")[1].split(".")[0]) for k in w
                                 if k.startswith(f"flow.flows.{self.flow_ids[0]}.enc.in_layers."))

    # ------------------------------------------------------------------ attention / encoder
    def attention(self, p: str, x: torch.Tensor, mask2d: torch.Tensor) -> torch.Tensor:
        """reference attentions.py:216-257 with the relative terms in banded form.

        s_ij = q_i.k_j/sqrt(dk) + [|j-i|<=W] q_i.Ek[j-i+W]/sqrt(dk); masked -> -1e4; p = softmax_j;
        o_i = sum_j p_ij v_j + sum_{|j-i|<=W} p_ij Ev[j-i+W]."""
        w = self.w
        B, C, L = x.shape
        H, dk, W = self.n_heads, C // self.n_heads, self.window
        q = F.conv1d(x, w[p + ".conv_q.weight"], w[p + ".conv_q.bias"])
        k = F.conv1d(x, w[p + ".conv_k.weight"], w[p + ".conv_k.bias"])
        v = F.conv1d(x, w[p + ".conv_v.weight"], w[p + ".conv_v.bias"])
        q = q.view(B, H, dk, L).transpose(2, 3) / math.sqrt(dk)
        k = k.view(B, H, dk, L).transpose(2, 3)
        v = v.view(B, H, dk, L).transpose(2, 3)
        scores = q @ k.transpose(-2, -1)                               # [B,H,L,L]
        ek, ev = w[p + ".emb_rel_k"][0], w[p + ".emb_rel_v"][0]        # [2W+1, dk]
        i = torch.arange(L)
        rel = i[None, :] - i[:, None] + W                              # j - i + W
        band = (rel >= 0) & (rel <= 2 * W)
        relc = rel.clamp(0, 2 * W)
        rl = q @ ek.t()                                                # [B,H,L,2W+1]
        scores = scores + torch.gather(rl, 3, relc.expand(B, H, L, L)) * band
        scores = scores.masked_fill(~mask2d[:, None], -1e4)
        pa = F.softmax(scores, dim=-1)
        out = pa @ v
        pw = torch.zeros(B, H, L, 2 * W + 1, dtype=x.dtype)
        pw.scatter_add_(3, relc.expand(B, H, L, L), pa * band)
        out = out + pw @ ev
        out = out.transpose(2, 3).reshape(B, C, L)
        return F.conv1d(out, w[p + ".conv_o.weight"], w[p + ".conv_o.bias"])

    def encoder(self, p: str, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """reference attentions.py:32-44 (post-LN, conv-FFN with 'same' padding :377-403)."""
        w = self.w
        mf = mask.to(x.dtype)[:, None, :]
        mask2d = mask[:, :, None] & mask[:, None, :]
        k = self.kernel_size
        pl, pr = (k - 1) // 2, k // 2
        x = x * mf
        for i in range(self.n_enc[p]):
            y = self.attention(f"{p}.attn_layers.{i}", x, mask2d)
            x = channel_layer_norm(x + y, w[f"{p}.norm_layers_1.{i}.gamma"], w[f"{p}.norm_layers_1.{i}.beta"])
            f = f"{p}.ffn_layers.{i}"
            y = F.conv1d(F.pad(x * mf, (pl, pr)), w[f + ".conv_1.weight"], w[f + ".conv_1.bias"])
            y = torch.relu(y)
            y = F.conv1d(F.pad(y * mf, (pl, pr)), w[f + ".conv_2.weight"], w[f + ".conv_2.bias"]) * mf
            x = channel_layer_norm(x + y, w[f"{p}.norm_layers_2.{i}.gamma"], w[f"{p}.norm_layers_2.{i}.beta"])
        return x * mf

    # ------------------------------------------------------------------ variance adaptor
    def duration_predictor(self, x, mf, g):
        """reference models.py:125-139"""
        w, p = self.w, "duration_predictor"
        if g is not None:
            x = x + F.conv1d(g, w[p + ".cond.weight"], w[p + ".cond.bias"])
        x = torch.relu(F.conv1d(x * mf, w[p + ".conv_1.weight"], w[p + ".conv_1.bias"], padding=1))
        x = channel_layer_norm(x, w[p + ".norm_1.gamma"], w[p + ".norm_1.beta"])
        x = torch.relu(F.conv1d(x * mf, w[p + ".conv_2.weight"], w[p + ".conv_2.bias"], padding=1))
        x = channel_layer_norm(x, w[p + ".norm_2.gamma"], w[p + ".norm_2.beta"])
        return F.conv1d(x * mf, w[p + ".proj.weight"], w[p + ".proj.bias"]) * mf

    def pitch_predictor(self, x, mask, g):
        """reference models.py:547-556"""
        w, p = self.w, "pitch_predictor"
        mf = mask.to(x.dtype)[:, None, :]
        if g is not None:
            x = x + F.conv1d(g, w[p + ".cond.weight"], w[p + ".cond.bias"])
        x = self.encoder(p + ".pitch_net", x * mf, mask) * mf
        return F.conv1d(x, w[p + ".proj_f0.weight"], w[p + ".proj_f0.bias"]).squeeze(1)

    def energy_predictor(self, x, g):
        """reference frame_prior_network.py:104-109, 124-129: no masking anywhere."""
        w, p = self.w, "energy_predictor"
        q = p + ".predictor.conv_layer"
        if g is None:       # the reference calls torch.detach(None) here (frame_prior_network.py:125)
            raise TypeError("energy_predictor needs the speaker embedding: a speaker-less model runs only with an "
                            "energy_control tensor")
        x = x + F.conv1d(g, w[p + ".cond.weight"], w[p + ".cond.bias"])
        x = torch.relu(F.conv1d(x, w[q + ".conv_1.conv.weight"], w[q + ".conv_1.conv.bias"], padding=1))
        x = channel_layer_norm(x, w[q + ".layer_norm_1.weight"], w[q + ".layer_norm_1.bias"])
        x = torch.relu(F.conv1d(x, w[q + ".conv_2.conv.weight"], w[q + ".conv_2.conv.bias"], padding=1))
        x = channel_layer_norm(x, w[q + ".layer_norm_2.weight"], w[q + ".layer_norm_2.bias"])
        lw, lb = w[p + ".predictor.linear_layer.weight"], w[p + ".predictor.linear_layer.bias"]
        return F.conv1d(x, lw[:, :, None], lb).squeeze(1)

    @staticmethod
    def length_regulate(x: torch.Tensor, duration: torch.Tensor):
        """reference models.py:426-455: phoneme p of row b repeated max(int(dur[b,p]), 0) times, zero-padded."""
        B, C, P = x.shape
        d = duration.reshape(B, -1)[:, :P]
        d = d.to(torch.float64).trunc().clamp_min(0).to(torch.int64) if d.is_floating_point() else d.clamp_min(0).to(torch.int64)
        lens = d.sum(1)
        T = int(lens.max())
        out = torch.zeros(B, C, T, dtype=torch.float32)
        for b in range(B):
            out[b, :, :int(lens[b])] = torch.repeat_interleave(x[b], d[b], dim=1)
        return out, lens

    # ------------------------------------------------------------------ flow
    def wn(self, p: str, x, mf, g):
        """reference modules.py:144-172 with commons.py:98-105 (tanh * sigmoid gate)."""
        w = self.w
        H = x.shape[1]
        out = torch.zeros_like(x)
        gc = None if g is None else F.conv1d(g, w[p + ".cond_layer.weight"], w[p + ".cond_layer.bias"])
        n = 0
        while f"{p}.in_layers.{n}.weight" in w:                 # 4 layers in a flow, 16 in the posterior encoder
            n += 1
        for i in range(n):
            wi = w[f"{p}.in_layers.{i}.weight"]
            x_in = F.conv1d(x, wi, w[f"{p}.in_layers.{i}.bias"], padding=(wi.shape[2] - 1) // 2)
            a = x_in if gc is None else x_in + gc[:, i * 2 * H:(i + 1) * 2 * H]
            acts = torch.tanh(a[:, :H]) * torch.sigmoid(a[:, H:])
            rs = F.conv1d(acts, w[f"{p}.res_skip_layers.{i}.weight"], w[f"{p}.res_skip_layers.{i}.bias"])
            if i < n - 1:
                x = (x + rs[:, :H]) * mf
                out = out + rs[:, H:]
            else:
                out = out + rs
        return out * mf

    def flow_reverse(self, z, mf, g):
        """reference models.py:213-216 (reversed: Flip then coupling) and modules.py:320-339 (mean_only)."""
        w = self.w
        half = z.shape[1] // 2
        for fid in reversed(self.flow_ids):
            z = torch.flip(z, [1])
            p = f"flow.flows.{fid}"
            x0, x1 = z[:, :half], z[:, half:]
            h = F.conv1d(x0, w[p + ".pre.weight"], w[p + ".pre.bias"]) * mf
            h = self.wn(p + ".enc", h, mf, g)
            m = F.conv1d(h, w[p + ".post.weight"], w[p + ".post.bias"]) * mf
            z = torch.cat([x0, (x1 - m) * mf], 1)
        return z

    def flow_forward(self, z, mf, g):
        """reference models.py:209-212 (coupling then Flip, layers in order) and modules.py:320-335 (mean_only)."""
        w = self.w
        half = z.shape[1] // 2
        for fid in self.flow_ids:
            p = f"flow.flows.{fid}"
            x0, x1 = z[:, :half], z[:, half:]
            h = F.conv1d(x0, w[p + ".pre.weight"], w[p + ".pre.bias"]) * mf
            h = self.wn(p + ".enc", h, mf, g)
            m = F.conv1d(h, w[p + ".post.weight"], w[p + ".post.bias"]) * mf
            z = torch.flip(torch.cat([x0, m + x1 * mf], 1), [1])
        return z

    # ------------------------------------------------------------------ voice conversion
    def posterior_encoder(self, y, mf, g, noise):
        """reference models.py:243-250: pre (1x1) -> WN (16 layers) -> proj; z = (m + eps * exp(logs)) * mask."""
        w = self.w
        x = F.conv1d(y, w["enc_q.pre.weight"], w["enc_q.pre.bias"]) * mf
        x = self.wn("enc_q.enc", x, mf, g)
        stats = F.conv1d(x, w["enc_q.proj.weight"], w["enc_q.proj.bias"]) * mf
        m, logs = torch.chunk(stats, 2, dim=1)
        return (m + noise * torch.exp(logs)) * mf, m, logs

    @torch.no_grad()
    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt, noise: Optional[torch.Tensor] = None):
        """reference models.py:769-777.  y: linear spectrogram [B, spec_channels, T]; ``noise`` replaces the
        torch.randn_like draw of the posterior encoder (models.py:249)."""
        w = self.w
        g_src = w["emb_g.weight"][sid_src].unsqueeze(-1)
        g_tgt = w["emb_g.weight"][sid_tgt].unsqueeze(-1)
        y = y.to(torch.float32)
        y_mask = sequence_mask(y_lengths, y.shape[2]).unsqueeze(1)
        mf = y_mask.to(torch.float32)
        if noise is None:
            noise = torch.randn(y.shape[0], w["enc_q.proj.weight"].shape[0] // 2, y.shape[2])
        z, m_q, logs_q = self.posterior_encoder(y, mf, g_src, noise)
        z_p = self.flow_forward(z, mf, g_src)
        z_hat = self.flow_reverse(z_p, mf, g_tgt)
        o_hat = self.generator(z_hat * mf, g_tgt)
        return o_hat, mf, (z, z_p, z_hat)

    # ------------------------------------------------------------------ generator
    def generator(self, x, g):
        """reference models.py:284-303 and modules.py:206-219 (ResBlock1)."""
        w = self.w
        if self.generator_bf16:
            def q(t):
                return t.to(torch.bfloat16).to(torch.float32)
        else:
            def q(t):
                return t
        x = F.conv1d(q(x), q(w["dec.conv_pre.weight"]), w["dec.conv_pre.bias"], padding=3)
        if g is not None:
            x = x + F.conv1d(g, w["dec.cond.weight"], w["dec.cond.bias"])
        nk = len(self.rb_k)
        for i, (u, ku) in enumerate(self.ups):
            x = F.leaky_relu(x, LRELU_SLOPE)
            x = F.conv_transpose1d(q(x), q(w[f"dec.ups.{i}.weight"]), w[f"dec.ups.{i}.bias"], stride=u,
                                   padding=(ku - u) // 2)
            xs = None
            for j in range(nk):
                r = i * nk + j
                k = self.rb_k[j]
                y = x
                for n, d in enumerate(self.rb_d[j]):
                    t = F.leaky_relu(y, LRELU_SLOPE)
                    if self.resblock == "2":             # modules.ResBlock2 (reference modules.py:241-250): x = c(lrelu(x)) + x
                        t = F.conv1d(q(t), q(w[f"dec.resblocks.{r}.convs.{n}.weight"]),
                                     w[f"dec.resblocks.{r}.convs.{n}.bias"], padding=(k * d - d) // 2, dilation=d)
                        y = t + y
                        continue
                    t = F.conv1d(q(t), q(w[f"dec.resblocks.{r}.convs1.{n}.weight"]),
                                 w[f"dec.resblocks.{r}.convs1.{n}.bias"], padding=(k * d - d) // 2, dilation=d)
                    t = F.leaky_relu(t, LRELU_SLOPE)
                    t = F.conv1d(q(t), q(w[f"dec.resblocks.{r}.convs2.{n}.weight"]),
                                 w[f"dec.resblocks.{r}.convs2.{n}.bias"], padding=(k - 1) // 2)
                    y = t + y
                xs = y if xs is None else xs + y
            x = xs / nk
        x = F.leaky_relu(x)                      # default slope 0.01 (reference models.py:299)
        x = F.conv1d(x, w["dec.conv_post.weight"], None, padding=3)
        return torch.tanh(x)

    # ------------------------------------------------------------------ infer
    @torch.no_grad()
    def infer(self, phonemes, phonemes_lengths, sid=None, noise_scale=1, max_len=None, energy_control=None,
              pitch_control=None, duration_control=None, noise: Optional[torch.Tensor] = None,
              stages: Optional[dict] = None):
        """reference models.py:717-767.  ``noise`` replaces the torch.randn_like draw (:763) when given;
        ``stages`` (a dict) receives every stage-boundary tensor."""
        w = self.w
        st = stages if stages is not None else {}
        g = w["emb_g.weight"][sid].unsqueeze(-1) if self.n_speakers > 0 else None    # :719-722
        # text encoder, models.py:175-181
        C = w["enc_p.symbol_emb.weight"].shape[1]
        B, P = phonemes.shape
        pmask = sequence_mask(phonemes_lengths, P)
        mf = pmask.to(torch.float32)[:, None, :]
        x = (w["enc_p.symbol_emb.weight"][phonemes] * math.sqrt(C)).transpose(1, 2)
        x = self.encoder("enc_p.encoder", x * mf, pmask)
        st["enc_p"] = x.clone()
        # duration, :726-733
        if isinstance(duration_control, torch.Tensor):
            duration = duration_control
        else:
            dc = 1 if duration_control is None else duration_control
            logw = self.duration_predictor(x, mf, g)
            st["logw"] = logw
            duration = torch.ceil((torch.exp(logw) * mf - 1) * dc)
        # pitch, :736-743
        if isinstance(pitch_control, torch.Tensor):
            lf0 = (2595. * torch.log10(1. + pitch_control / 700.)) / 500
        else:
            pc = 1 if pitch_control is None else pitch_control
            lf0 = self.pitch_predictor(x, pmask, g) * pc
        st["lf0"] = lf0
        x = x + F.conv1d(lf0.unsqueeze(1), w["pitch_prenet.weight"], w["pitch_prenet.bias"], padding=1)
        f0 = (torch.pow(10, lf0 * 500 / 2590) - 1) * 700                          # 2590: reference quirk
        # energy, :746-753
        if isinstance(energy_control, torch.Tensor):
            ne = (energy_control - 60) / 36
        else:
            ec = 1 if energy_control is None else energy_control
            ne = (((self.energy_predictor(x, g) * 36 + 60) * ec) - 60) / 36
        st["norm_energy"] = ne
        x = x + F.conv1d(ne.unsqueeze(1), w["energy_prenet.weight"], w["energy_prenet.bias"], padding=1)
        energy = ne * 36 + 60
        st["x_phone"] = x
        # length regulator, :756-758
        x_frame, x_lengths = self.length_regulate(x, duration)
        T = x_frame.shape[2]
        fmask = sequence_mask(x_lengths, T)
        fm = fmask.to(torch.float32)[:, None, :]
        st["x_frame"], st["x_lengths"] = x_frame, x_lengths
        # frame prior + projection, :760-762
        h = self.encoder("frame_prior_net.fft_block", x_frame * fm, fmask)
        st["prior"] = h
        stats = F.conv1d(h, w["project.proj.weight"], w["project.proj.bias"]) * fm
        m_p, logs_p = stats[:, :C], stats[:, C:]
        # sample, :763
        eps = torch.randn_like(m_p) if noise is None else noise
        z_p = m_p + eps * torch.exp(logs_p) * noise_scale
        # flow, :764
        z = self.flow_reverse(z_p, fm, g)
        # generator, :765
        o = self.generator((z * fm)[:, :, :max_len], g)
        return o, fmask.unsqueeze(1), (z, z_p, m_p, logs_p), duration, f0, energy
