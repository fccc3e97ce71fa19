"""Drop-in for the reference's ``models.SynthesizerTrn`` inference surface on MI355X.

Construction arguments, ``load_state_dict`` / checkpoint layout, the ``infer`` signature, its 6-tuple return and its
error behaviour follow reference models.py:577-767.  All arithmetic runs in libvits_amd.so (hand-written HIP
kernels, include/vits_amd.h); PyTorch is used here only to own device memory, to provide the current HIP stream and
to draw the prior noise when the caller does not pass one.  There is no fallback: without the library or without a
GPU this module raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import time
import threading
import warnings
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import _native as N
from . import synthetic, weights

_DUR_DTYPES = {torch.float32: 0, torch.int64: 1, torch.int32: 2, torch.float64: 3}


# VITS_READBACK_SPIN_US: how long _read_back polls for its copy before it blocks in the runtime (0: never polls).
# Batch-1 latency 3.41 -> 3.39 ms; nothing either way in throughput mode.
_SPIN_US = float(os.environ.get("VITS_READBACK_SPIN_US", "2000"))


def _rccl_group_live() -> bool:
    """True when this process already holds a torch.distributed group on the nccl (= RCCL) backend."""
    try:
        import torch.distributed as dist
        return bool(dist.is_available() and dist.is_initialized() and "nccl" in str(dist.get_backend()))
    except Exception:           # no distributed support in this torch build: no communicator either
        return False


class SynthesizerTrn:
    """Inference-only synthesizer: infer, infer_stream and voice_conversion (the posterior encoder is built when the
    checkpoint holds its weights; the reference's training forward and discriminators are not)."""

    # The two streams of throughput mode (front end, generator) belong to the process, one pair per device, shared by
    # every model: the device schedules four hardware queues at once, and a pair per model would put a second model's
    # pipeline onto queues the first one's streams already hold (measured: 6.7 % slower).
    _PIPELINE: Dict[int, tuple] = {}
    _COMM_FIRST: Dict[int, bool] = {}

    def __init__(self, n_vocab, spec_channels, hop_length, sampling_rate, segment_size, inter_channels,
                 hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, resblock,
                 resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates, upsample_initial_channel,
                 upsample_kernel_sizes, n_speakers=0, gin_channels=0, use_sdp=False, freeze_textencoder=False,
                 freeze_decoder=False, **kwargs):
        if str(resblock) not in ("1", "2"):
            raise ValueError("resblock must be '1' (modules.ResBlock1) or '2' (modules.ResBlock2)")
        # n_speakers == 0: speaker-less, g = None everywhere (reference models.py:719-722).  n_speakers == 1 is a
        # configuration the reference cannot run itself: no embedding table is created for it (models.py:660) and a
        # speaker is looked up all the same (:719).
        if n_speakers < 0 or n_speakers == 1 or (n_speakers > 1 and gin_channels <= 0):
            raise NotImplementedError("n_speakers must be 0 (speaker-less) or > 1 with gin_channels > 0")
        # arithmetic of the conv kernels.  fp32-grade: "f16x2" (default: the generator's convs as three fp16 products
        # of two-term operand splits, everything else bf16x3), "bf16x3" (six bf16 products of three-term splits; full
        # fp32 range) and "fp32" (fp32 MFMA).  "bf16": generator convs in plain bf16, fp32 accumulate.
        # Keyword `precision` or env VITS_PRECISION
        prec = kwargs.pop("precision", None) or os.environ.get("VITS_PRECISION", "f16x2")
        if prec not in N.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(N.PRECISIONS)}")
        self.precision = prec
        self.cfg = synthetic.ModelConfig(
            n_vocab=n_vocab, spec_channels=spec_channels, hop_length=hop_length, sampling_rate=sampling_rate,
            segment_size=segment_size, inter_channels=inter_channels, hidden_channels=hidden_channels,
            filter_channels=filter_channels, n_heads=n_heads, n_layers=n_layers, kernel_size=kernel_size,
            p_dropout=p_dropout, resblock=str(resblock), resblock_kernel_sizes=tuple(resblock_kernel_sizes),
            resblock_dilation_sizes=tuple(tuple(d) for d in resblock_dilation_sizes),
            upsample_rates=tuple(upsample_rates), upsample_initial_channel=upsample_initial_channel,
            upsample_kernel_sizes=tuple(upsample_kernel_sizes), n_speakers=n_speakers, gin_channels=gin_channels)
        self.n_vocab, self.n_speakers, self.gin_channels = n_vocab, n_speakers, gin_channels
        self.hidden_channels, self.inter_channels = hidden_channels, inter_channels
        self.hop_length, self.sampling_rate = hop_length, sampling_rate
        self.total_upsample = 1
        for u in upsample_rates:
            self.total_upsample *= u
        self._manifest = synthetic.state_dict_manifest(self.cfg)
        self._sd: Dict[str, torch.Tensor] = {}
        self._handle = C.c_void_p()
        self._device_tensors = []           # keeps packed weights alive for the library
        self._ws: Optional[torch.Tensor] = None
        self._ready = False
        self.device = torch.device("cuda", 0)
        # One workspace and one set of generator fork / join events serve every call, so the model is neither
        # re-entrant nor multi-stream: a second thread inside a call gets a RuntimeError (the reference's server
        # serialises requests the same way, inference_api.py:13,40), and a call on another stream than the previous
        # one first waits for the previous call's work (_busy / _last_use below).
        self._busy = threading.Lock()
        self._last_use = None                # (stream, event recorded on it after the last call)
        self._range = None                   # device int32: the f16x2 kernels raise it when they clamp an activation
        self._range_warned = False
        self.strict_range = bool(kwargs.pop("strict_range", False)) or os.environ.get("VITS_STRICT_RANGE") == "1"
        c = N.VitsConfig()
        cfg = self.cfg
        c.n_vocab, c.hidden_channels, c.inter_channels = cfg.n_vocab, cfg.hidden_channels, cfg.inter_channels
        c.filter_channels, c.n_heads, c.n_layers, c.kernel_size = cfg.filter_channels, cfg.n_heads, cfg.n_layers, cfg.kernel_size
        c.gin_channels, c.n_speakers = cfg.gin_channels, cfg.n_speakers
        c.pitch_layers, c.dur_filter, c.energy_filter = cfg.pitch_layers, cfg.dur_filter, cfg.energy_filter
        c.n_flows, c.flow_layers, c.flow_kernel, c.window_size = cfg.n_flows, cfg.flow_layers, cfg.flow_kernel, cfg.window_size
        c.n_upsamples = len(cfg.upsample_rates)
        for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            c.upsample_rates[i], c.upsample_kernel_sizes[i] = u, k
        c.upsample_initial_channel = cfg.upsample_initial_channel
        c.resblock = int(cfg.resblock)
        c.n_resblock_kernels = len(cfg.resblock_kernel_sizes)
        for j, k in enumerate(cfg.resblock_kernel_sizes):
            c.resblock_kernel_sizes[j] = k
        nd = {len(d) for d in cfg.resblock_dilation_sizes}
        if len(nd) != 1:
            raise NotImplementedError("every resblock must have the same number of dilations")
        c.n_resblock_dilations = nd.pop()
        for j, dl in enumerate(cfg.resblock_dilation_sizes):
            for q, d in enumerate(dl):
                c.resblock_dilation_sizes[j][q] = d
        N.check(N.lib.vits_model_create(C.byref(c), C.byref(self._handle)), "vits_model_create")

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value and N is not None and getattr(N, "lib", None) is not None:   # not at interpreter exit
            N.lib.vits_model_destroy(h)
            self._handle = C.c_void_p()

    # ---- torch.nn.Module look-alikes used by the reference's callers (inference.py:23-42) ----
    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("vits_amd.SynthesizerTrn is inference-only")
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("vits_amd runs on an MI355X only; there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device != self.device:
            self.device = device
            if self._sd or getattr(self, "_packed", None) is not None:
                self._upload()
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", 0 if device is None else device))

    def state_dict(self):
        """Every checkpoint key with the tensor currently held, or a shape-only (meta) tensor when none is."""
        return OrderedDict((k, self._sd.get(k, torch.empty(shp, device="meta"))) for k, shp in self._manifest.items())

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts the reference's generator state dict.  Keys infer never reads (enc_q.*, enc_p.proj.*, ...) are
        optional; any other absent key is an error, because an uninitialised weight cannot produce meaningful audio."""
        sd = {}
        for k, v in state_dict.items():
            if isinstance(v, torch.Tensor) and v.device.type != "meta":
                if k in self._manifest and tuple(v.shape) != tuple(self._manifest[k]):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)}, model {tuple(self._manifest[k])}")
                sd[k] = v.detach().to("cpu", torch.float32)
        unused = weights.UNUSED_PREFIXES
        if self.n_speakers == 0:                 # conditioning convs a speaker-less checkpoint may hold, never applied
            unused = unused + tuple(k for k in self._manifest if ".cond." in k or ".cond_layer." in k)
        needed = [k for k in self._manifest if not k.startswith(unused)]
        missing = [k for k in needed if k not in sd]
        if missing:
            raise RuntimeError(f"state dict is missing {len(missing)} tensors the synthesis path needs, e.g. {missing[:4]}")
        if strict:
            unexpected = [k for k in sd if k not in self._manifest]
            if unexpected:
                raise RuntimeError(f"unexpected keys in state dict: {unexpected[:4]}")
        self._sd = sd
        self._upload()
        return self

    @classmethod
    def from_packed(cls, path: str, device=None) -> "SynthesizerTrn":
        """Build a model from a file written by vits_amd.packed.save_packed (or `python -m vits_amd.packed`): the
        configuration, the arithmetic mode and every folded, fragment-ordered tensor come from the file and go to the
        GPU as they are.  state_dict() of such a model holds shapes only."""
        from . import packed
        cfg, precision, pw = packed.load_packed(path)
        net = cls(cfg.n_vocab, cfg.spec_channels, cfg.hop_length, cfg.sampling_rate, cfg.segment_size,
                  precision=precision, **cfg.kwargs())
        if device is not None:
            net.device = torch.device(device)
            if net.device.index is None:
                net.device = torch.device("cuda", torch.cuda.current_device())
        net._packed = pw
        net._upload(pw)
        return net

    def _upload(self, pw=None):
        if not torch.cuda.is_available():
            raise RuntimeError("vits_amd needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
        if pw is None:
            pw = getattr(self, "_packed", None) if not self._sd else None
        if pw is None:
            pw = weights.prepare(self._sd, self.cfg, N.PRECISIONS[self.precision])
        self._device_tensors = []
        h = self._handle
        with torch.cuda.device(self.device):
            for name, t in pw.plain.items():
                d = t.to(self.device)
                self._device_tensors.append(d)
                N.check(N.lib.vits_model_set_tensor(h, name.encode(), d.data_ptr(), d.numel()), f"set_tensor {name}")
            for name, (w, b, cin, rows, taps, dil, mode, prec) in pw.convs.items():
                dw = w.to(self.device)
                db = None if b is None else b.to(self.device)
                self._device_tensors += [dw, db]
                N.check(N.lib.vits_model_set_conv(h, name.encode(), dw.data_ptr(), N.ptr(db), cin, rows, taps, dil, mode,
                                                  prec),
                        f"set_conv {name}")
            for name, w in getattr(pw, "alts", {}).items():
                dw = w.to(self.device)
                self._device_tensors.append(dw)
                N.check(N.lib.vits_model_set_conv_alt(h, name.encode(), dw.data_ptr()), f"set_conv_alt {name}")
            N.check(N.lib.vits_model_finalize(h), "vits_model_finalize")
            # two device status words that live as long as the model: [0] id-range errors of the embedding stages of
            # the call in progress (zero unless one was just raised), [1] the sticky fp16-range flag
            self._flags = torch.zeros(2, dtype=torch.int32, device=self.device)
            self._err, self._range = self._flags[0:1], self._flags[1:2]
            N.check(N.lib.vits_model_set_range_flag(h, self._range.data_ptr()), "set_range_flag")
            torch.cuda.synchronize(self.device)
        self._last_use = None
        self._ready = True

    # ---- single-stream, single-thread discipline ----
    @contextlib.contextmanager
    def _exclusive(self):
        """Entered by every call that uses the model's workspace.  Raises if another thread is inside one; makes the
        current stream wait for the previous call's work when that ran on a different stream."""
        if not self._busy.acquire(blocking=False):
            raise RuntimeError("vits_amd.SynthesizerTrn is not re-entrant: another thread is inside infer / "
                               "infer_batches / voice_conversion on this model (use one model per thread)")
        try:
            cur = torch.cuda.current_stream(self.device)
            if self._last_use is not None and self._last_use[0] != cur:
                cur.wait_event(self._last_use[1])
            yield
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._last_use = (torch.cuda.current_stream(self.device), ev)
        finally:
            self._busy.release()

    def _report_range(self, flag: int) -> None:
        if not flag:
            return
        msg = ("f16x2 generator kernels clamped activations to the fp16 range (|x| > 65504): this checkpoint's "
               "generator leaves the range the default arithmetic covers and the waveform is saturated there; "
               "build the model with precision=\"bf16x3\" (full fp32 range, same accuracy)")
        if self.strict_range:
            self._range.zero_()
            raise OverflowError(msg)
        if not self._range_warned:
            self._range_warned = True
            warnings.warn(msg, RuntimeWarning, stacklevel=3)

    def range_overflow(self) -> bool:
        """True if any f16x2 kernel has clamped an activation to the fp16 range since the model was loaded
        (synchronises the device).  infer itself reads the flag without an extra synchronisation, together with the
        frame counts of the NEXT call, and then warns once (or raises OverflowError with strict_range=True)."""
        if self._range is None:
            return False
        return bool(int(self._range.cpu()))

    # ---- hardware-queue budget ----
    def _budget_generator_streams(self, caller) -> int:
        """How many streams vits_generator may spread its ResBlock chains over for a call from `caller`, and tells the
        library.  A HIP process schedules four hardware queues at once, and every stream that has carried work holds
        one: the process's default stream, the caller's if it is another, and -- for the rest of the model's life,
        once infer_batches has run -- its front-end and generator streams, busy or idle.  With a fifth queue in play a
        host copy issued beside the generator's streams was seen to hang, so the chains get what the four leave over;
        a caller on a stream of its own is taken to hold one more stream besides it."""
        default = caller == torch.cuda.default_stream(self.device)
        used = (1 if default else 3) + (2 if self.device.index in SynthesizerTrn._PIPELINE else 0)
        n = 1 + max(0, min(2, 4 - used))
        n = int(os.environ.get("VITS_INFER_GEN_STREAMS", n))
        N.check(N.lib.vits_model_set_generator_streams(self._handle, n), "set_generator_streams")
        return n

    def _read_back(self, ints: torch.Tensor) -> torch.Tensor:
        """Device int32 vector -> host, through the model's pinned buffer: an asynchronous copy on the current stream
        followed by a wait for that stream (not a blocking copy to pageable memory, which waits inside the copy call
        with a barrier packet on the queue)."""
        n = ints.numel()
        if getattr(self, "_pin", None) is None or self._pin.numel() < n:
            self._pin = torch.empty(max(n, 1024), dtype=torch.int32).pin_memory()
        self._pin[:n].copy_(ints, non_blocking=True)
        stream = torch.cuda.current_stream(self.device)
        if _SPIN_US > 0:
            # the copy is a few microseconds behind the last kernel: poll an event for a bounded time before handing
            # the wait to the runtime (whose wake-up costs tens of microseconds of an otherwise idle GPU)
            ev = torch.cuda.Event()
            ev.record(stream)
            deadline = time.perf_counter() + _SPIN_US * 1e-6
            while not ev.query():
                if time.perf_counter() > deadline:
                    stream.synchronize()
                    break
        else:
            stream.synchronize()
        return self._pin[:n].clone()

    # ---- workspace ----
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != self.device:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- the hot path ----
    @torch.no_grad()
    def infer(self, phonemes, phonemes_lengths, sid=None, noise_scale=1, max_len=None, energy_control=None,
              pitch_control=None, duration_control=None, noise=None, stages: Optional[dict] = None):
        """reference models.py:717-767.  Extra keyword ``noise`` ([B, inter_channels, T]) replaces the
        torch.randn_like draw at :763; ``stages`` (a dict) receives stage-boundary tensors for tests."""
        dev = self.device
        with torch.cuda.device(dev), self._exclusive():
            fe = self._frontend(phonemes, phonemes_lengths, sid, noise_scale, max_len, energy_control, pitch_control,
                                duration_control, noise, stages)
            B, T, t_dec = fe["B"], fe["T"], fe["t_dec"]
            self._budget_generator_streams(torch.cuda.current_stream(dev))
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = self._workspace(N.lib.vits_generator_workspace_bytes(self._handle, B, t_dec))
            o = torch.empty(B, 1, t_dec * self.total_upsample, dtype=torch.float32, device=dev)
            N.check(N.lib.vits_generator(self._handle, fe["z"].data_ptr(), fe["totals"].data_ptr(), fe["cond"].data_ptr(),
                                         B, T, t_dec, o.data_ptr(), ws.data_ptr(), ws.numel(), stream), "generator")
        return (o,) + fe["outputs"]

    @torch.no_grad()
    def infer_batches(self, batches):
        """Throughput mode for a sequence of independent batches: yields infer()'s tuple for each element of
        `batches` (an iterable of dicts of infer()'s arguments), in order, with the values infer() returns: the same
        bytes for batches of fewer than 2048 frames in all; from there on the frame-rate convs of the front end run on
        their f16x2 packing here (they take less of the chip from the generator beside them) where a plain call
        keeps the latency-oriented bf16x3 kernel, and the waveforms agree to ~1e-6 instead (VITS_FE_F16X2_COLS moves
        the size, VITS_FE_F16X2=0 switches the packing off).  Within this mode a batch's bytes do not depend on the
        batches around it.

        `batches` is consumed lazily, one element ahead of the results: element i + 1 is taken only after the result
        of element i - 1 has been handed to the consumer, so a serving loop can feed requests as they arrive (the
        reference's server takes them one at a time, reference inference_api.py:38-60).  An element that is None
        means "nothing more right now": the results still in the pipeline are yielded before the next element is
        asked for.  Host tensors in an element go to the GPU through pinned memory with asynchronous copies on the
        front-end stream; no blocking copy is issued while the streams are in flight.

        The front end of a batch (text encoder ... flow: many short kernels and one host round trip for the frame
        count) leaves most of the GPU idle, and the generator of the previous batch does not depend on it, so the two
        run on two streams with separate workspaces: while batch i's generator saturates the chip, batch i+1's front
        end fills the gaps.  Results are handed over on the caller's current stream (it waits on the generator's
        event; no host synchronisation), one batch behind the submission.

        Streams: the caller's, the front-end stream, the generator stream and ONE library side stream for the
        generator's ResBlock chains -- four in all, one hardware queue each (a HIP process has four by default, and
        streams of different priority never share one).  With a fifth queue in play (two side streams) this mode was
        seen to deadlock, so the generator is held to one side stream while it runs, and every later call on this
        model keeps counting the two streams created here (_budget_generator_streams)."""
        dev = self.device
        if not self._busy.acquire(blocking=False):
            raise RuntimeError("vits_amd.SynthesizerTrn is not re-entrant: another call is in progress on this model")
        lib, h = N.lib, self._handle
        try:
            with torch.cuda.device(dev):
                if dev.index not in SynthesizerTrn._PIPELINE:
                    # the front end's kernels are short and few blocks wide: on a high-priority stream they take the
                    # slots the generator's blocks free up instead of queueing behind whole generator launches
                    prio = int(os.environ.get("VITS_FE_PRIORITY", "-1"))
                    SynthesizerTrn._PIPELINE[dev.index] = (torch.cuda.Stream(dev, priority=prio), torch.cuda.Stream(dev))
                    # a communicator set up before these streams already holds hardware queues of its own
                    SynthesizerTrn._COMM_FIRST[dev.index] = _rccl_group_live()
                if getattr(self, "_s_fe", None) is None or self._s_fe.device != dev:
                    self._s_fe, self._s_gen = SynthesizerTrn._PIPELINE[dev.index]
                    self._ws_gen = None
                caller = torch.cuda.current_stream(dev)
                if self._last_use is not None and self._last_use[0] != caller:
                    caller.wait_event(self._last_use[1])
                # Hardware queues: one per stream in use, the process's default stream included (normal-priority streams
                # share four, a high-priority stream has one of its own), and four are what the device schedules at
                # once.  Called on the default stream that is default + front end + generator + one side stream; a
                # caller on a stream of its own already holds the fourth, so the generator then stays on one stream.
                # An RCCL process group that was initialised before this model's streams existed took its queues first,
                # and the generator's side stream would then share one with another stream of this call (a C2 step
                # measured 15.1 ms instead of 11.5; 12.0 with the chains on one stream): set the model up and run one
                # infer_batches before init_process_group, and none of this applies.
                gen_streams = 2 if caller == torch.cuda.default_stream(dev) and not SynthesizerTrn._COMM_FIRST.get(dev.index) else 1
                N.check(lib.vits_model_set_generator_streams(h, int(os.environ.get("VITS_BATCHES_GEN_STREAMS", gen_streams))),
                        "set_generator_streams")
                # every front end of this call runs beside a generator (all but the first, and the first takes the same
                # kernels so that a batch's bytes do not depend on its place in the sequence)
                N.check(lib.vits_model_set_front_end_overlapped(h, 1), "set_front_end_overlapped")
                start = torch.cuda.Event()
                start.record(caller)
                self._s_fe.wait_event(start)                    # inputs prepared on the caller's stream are visible
                self._s_gen.wait_event(start)
                pending = None
                for kw in batches:
                    if kw is None:                              # the source has nothing right now: drain
                        if pending is not None:
                            yield self._hand_over(*pending, caller)
                            pending = None
                        continue
                    kw = dict(kw)
                    on_device = [v for v in kw.values() if isinstance(v, torch.Tensor) and v.is_cuda]
                    if on_device:
                        # made on the caller's stream, possibly after `start`: the front-end stream reads them
                        ready = torch.cuda.Event()
                        ready.record(caller)
                        self._s_fe.wait_event(ready)
                        for v in on_device:
                            v.record_stream(self._s_fe)
                    with torch.cuda.stream(self._s_fe):
                        for k, v in kw.items():
                            if isinstance(v, torch.Tensor) and not v.is_cuda:
                                kw[k] = v.pin_memory().to(dev, non_blocking=True)
                        args = (kw.pop("phonemes"), kw.pop("phonemes_lengths"))
                        fe = self._frontend(*args, kw.pop("sid", None), kw.pop("noise_scale", 1), kw.pop("max_len", None),
                                            kw.pop("energy_control", None), kw.pop("pitch_control", None),
                                            kw.pop("duration_control", None), kw.pop("noise", None), None)
                        if kw:
                            raise TypeError(f"unexpected arguments {sorted(kw)}")
                        fe_done = torch.cuda.Event()
                        fe_done.record(self._s_fe)
                    B, T, t_dec = fe["B"], fe["T"], fe["t_dec"]
                    if pending is not None:
                        # One generator call in flight at a time: the library forks and joins its ResBlock chains with
                        # one set of events per model, which must not be recorded again while the previous call still
                        # waits on them.  The front end above has already been enqueued (and run) beside that call.
                        pending[2].synchronize()
                    # The front end is complete before its generator is enqueued (a host wait: the generator stream
                    # then holds no barrier on another queue's marker).  The previous generator is still running or
                    # just finished, so the chip stays busy.
                    fe_done.synchronize()
                    with torch.cuda.stream(self._s_gen):
                        need = lib.vits_generator_workspace_bytes(h, B, t_dec)
                        if self._ws_gen is None or self._ws_gen.numel() < need:
                            self._ws_gen = None
                            self._ws_gen = torch.empty(int(need), dtype=torch.uint8, device=dev)
                        o = torch.empty(B, 1, t_dec * self.total_upsample, dtype=torch.float32, device=dev)
                        for t in (fe["z"], fe["totals"], fe["cond"]):
                            t.record_stream(self._s_gen)         # allocated on the front-end stream, read here
                        N.check(lib.vits_generator(h, fe["z"].data_ptr(), fe["totals"].data_ptr(), fe["cond"].data_ptr(),
                                                   B, T, t_dec, o.data_ptr(), self._ws_gen.data_ptr(),
                                                   self._ws_gen.numel(), self._s_gen.cuda_stream), "generator")
                        gen_done = torch.cuda.Event()
                        gen_done.record(self._s_gen)
                    if pending is not None:
                        yield self._hand_over(*pending, caller)
                    pending = (o, fe, gen_done)
                if pending is not None:
                    yield self._hand_over(*pending, caller)
        finally:
            # whatever ends the loop (exhaustion, an exception, the consumer dropping the generator): the next call on
            # this model starts after both side streams have drained.  The two streams stay (torch hands them out of
            # a pool and never destroys them), so later calls budget their generator streams around them.
            lib.vits_model_set_front_end_overlapped(h, 0)
            if getattr(self, "_s_fe", None) is not None:
                ev = torch.cuda.Event()
                self._s_gen.wait_stream(self._s_fe)
                ev.record(self._s_gen)
                self._last_use = (self._s_gen, ev)
            self._busy.release()

    @staticmethod
    def _hand_over(o, fe, gen_done, caller):
        caller.wait_event(gen_done)
        o.record_stream(caller)
        for t in fe["outputs"]:
            for u in (t if isinstance(t, tuple) else (t,)):
                if isinstance(u, torch.Tensor) and u.is_cuda:
                    u.record_stream(caller)
        return (o,) + fe["outputs"]

    @torch.no_grad()
    def infer_stream(self, phonemes, phonemes_lengths, sid=None, noise_scale=1, max_len=None, energy_control=None,
                     pitch_control=None, duration_control=None, noise=None, chunk_frames: int = 64):
        """Chunked synthesis: everything up to the latent z runs once, then the generator runs on windows of
        ``chunk_frames`` frames plus GENERATOR_HALO_FRAMES on each side and only the window's own samples are kept.
        The halo covers the generator's receptive field (13 frames a side for the reference configuration), so the
        chunks concatenate to exactly the waveform infer() returns, while the first audio is ready after one window
        instead of the whole utterance.  The reference's callers synthesise whole utterances only
        (reference inference_api.py:38-60).

        Returns an InferStream: iterate it for (first_sample, wav[B, 1, n]) pairs; its ``outputs`` attribute holds the
        rest of infer()'s tuple (x_mask, (z, z_p, m_p, logs_p), duration, F0, energy)."""
        if chunk_frames <= 0:
            raise ValueError("chunk_frames must be positive")
        with torch.cuda.device(self.device), self._exclusive():
            fe = self._frontend(phonemes, phonemes_lengths, sid, noise_scale, max_len, energy_control, pitch_control,
                                duration_control, noise, None)
        return InferStream(self, fe, int(chunk_frames))

    @torch.no_grad()
    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt, noise=None):
        """reference models.py:769-777: posterior encoder on the linear spectrogram y [B, spec_channels, T] with the
        source speaker, flow forward (source), flow reverse (target), generator (target).  Returns
        (o_hat [B, 1, T * hop], y_mask float [B, 1, T], (z, z_p, z_hat)).  Needs the checkpoint's enc_q.* weights.
        Extra keyword ``noise`` ([B, inter_channels, T]) replaces the torch.randn_like draw at models.py:249."""
        if not self._ready:
            raise RuntimeError("no weights loaded: call load_state_dict / utils.load_checkpoint first")
        lib, h, dev = N.lib, self._handle, self.device
        if self.n_speakers == 0:
            raise RuntimeError("voice_conversion needs a speaker-conditioned model")
        pc = lib.vits_posterior_channels(h)
        if pc < 0:
            raise RuntimeError("voice_conversion needs the posterior encoder: load a checkpoint that holds enc_q.*")
        if y.dim() != 3 or y.shape[1] != self.cfg.spec_channels:
            raise ValueError(f"y must be [B, {self.cfg.spec_channels}, T]")
        B, S, T = y.shape
        if T > 16384:
            raise ValueError("at most 16384 frames per utterance")
        IC = self.cfg.inter_channels
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev), self._exclusive():
            stream = torch.cuda.current_stream(dev).cuda_stream
            spec = torch.zeros(B, pc, T, **f32)                  # channel count rounded up to a multiple of 16
            spec[:, :S] = y.to(dev, torch.float32)
            lens = y_lengths.to(dev, torch.int32).contiguous()
            if lens.numel() != B:
                raise ValueError("y_lengths must have one entry per batch row")
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            conds = []
            for sid in (sid_src, sid_tgt):
                sids = sid.to(dev, torch.int64).contiguous().reshape(-1)
                if sids.numel() != B:
                    raise ValueError("sid_src and sid_tgt must have one entry per batch row")
                c = torch.empty(B, lib.vits_cond_rows(h), **f32)
                N.check(lib.vits_speaker_cond(h, sids.data_ptr(), B, c.data_ptr(), err.data_ptr(), stream), "speaker_cond")
                conds.append(c)
            if noise is None:
                noise_t = torch.randn(B, IC, T, **f32)
            else:
                noise_t = noise.to(dev, torch.float32).contiguous()
                if tuple(noise_t.shape) != (B, IC, T):
                    raise ValueError(f"noise must be [{B}, {IC}, {T}]")
            ws = self._workspace(max(lib.vits_posterior_workspace_bytes(h, B, T), lib.vits_flow_workspace_bytes(h, B, T),
                                     lib.vits_generator_workspace_bytes(h, B, T)))
            stats = torch.empty(B, 2 * IC, T, **f32)
            z = torch.empty(B, IC, T, **f32)
            N.check(lib.vits_posterior_encoder(h, spec.data_ptr(), lens.data_ptr(), conds[0].data_ptr(), B, T,
                                               noise_t.data_ptr(), stats.data_ptr(), z.data_ptr(), ws.data_ptr(),
                                               ws.numel(), stream), "posterior_encoder")
            z_p = z.clone()
            N.check(lib.vits_flow_forward(h, z_p.data_ptr(), lens.data_ptr(), conds[0].data_ptr(), B, T, ws.data_ptr(),
                                          ws.numel(), stream), "flow_forward")
            z_hat = z_p.clone()
            N.check(lib.vits_flow_reverse(h, z_hat.data_ptr(), lens.data_ptr(), conds[1].data_ptr(), B, T, ws.data_ptr(),
                                          ws.numel(), stream), "flow_reverse")
            o = torch.empty(B, 1, T * self.total_upsample, **f32)
            self._budget_generator_streams(torch.cuda.current_stream(dev))
            N.check(lib.vits_generator(h, z_hat.data_ptr(), lens.data_ptr(), conds[1].data_ptr(), B, T, T, o.data_ptr(),
                                       ws.data_ptr(), ws.numel(), stream), "generator")
            if int(err.cpu()) & 2:
                raise IndexError("speaker id out of range")
            y_mask = (torch.arange(T, device=dev)[None, :] < lens[:, None]).unsqueeze(1).to(torch.float32)
        return o, y_mask, (z, z_p, z_hat)

    def _frontend(self, phonemes, phonemes_lengths, sid, noise_scale, max_len, energy_control, pitch_control,
                  duration_control, noise, stages):
        """Everything of infer() before the generator: reference models.py:717-763."""
        if not self._ready:
            raise RuntimeError("no weights loaded: call load_state_dict / utils.load_checkpoint first")
        speakers = self.n_speakers > 0
        if speakers and sid is None:
            # the reference calls self.emb_g(None) here and dies with a TypeError (models.py:719-720)
            raise TypeError("sid is required: the model is speaker-conditioned (n_speakers > 0)")
        if not speakers and not isinstance(energy_control, torch.Tensor):
            # g is None, and the reference's energy predictor calls torch.detach(g) (frame_prior_network.py:125)
            raise TypeError("a speaker-less model (n_speakers == 0) predicts no energy: pass energy_control as a tensor")
        lib, h, dev = N.lib, self._handle, self.device
        cfg = self.cfg
        Cc, IC = cfg.hidden_channels, cfg.inter_channels
        if phonemes.dim() != 2:
            raise ValueError("phonemes must be [B, P]")
        B, P = phonemes.shape
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            ph = phonemes.to(dev, torch.int64).contiguous()
            lens = phonemes_lengths.to(dev, torch.int32).contiguous()
            if lens.numel() != B:
                raise ValueError("phonemes_lengths must have one entry per batch row")
            err = self._err                     # zero on entry: reset whenever it is reported, or was never read
            if getattr(self, "_err_pending", False):
                err.zero_()                     # the previous call ended before it read the word back
            self._err_pending = True
            if speakers:
                sids = sid.to(dev, torch.int64).contiguous().reshape(-1)
                if sids.numel() != B:
                    raise ValueError("sid must have one entry per batch row")
                cond = torch.empty(B, lib.vits_cond_rows(h), **f32)
                N.check(lib.vits_speaker_cond(h, sids.data_ptr(), B, cond.data_ptr(), err.data_ptr(), stream), "speaker_cond")
            else:                                # g = None: no conditioning conv is applied anywhere (sid is ignored)
                cond = torch.zeros(B, lib.vits_cond_rows(h), **f32)
            # text encoder
            x = torch.empty(B, Cc, P, **f32)
            ws_bytes = max(lib.vits_encoder_workspace_bytes(h, B, P), lib.vits_variance_workspace_bytes(h, B, P))
            ws = self._workspace(ws_bytes)
            N.check(lib.vits_text_encoder(h, ph.data_ptr(), lens.data_ptr(), B, P, x.data_ptr(), err.data_ptr(),
                                          ws.data_ptr(), ws.numel(), stream), "text_encoder")
            if stages is not None:
                stages["enc_p"] = x.clone()
            # variance adaptor
            dur_is_tensor = isinstance(duration_control, torch.Tensor)
            pitch_is_tensor = isinstance(pitch_control, torch.Tensor)
            energy_is_tensor = isinstance(energy_control, torch.Tensor)
            dur_scale = 1.0 if duration_control is None or dur_is_tensor else float(duration_control)
            pitch_scale = 1.0 if pitch_control is None or pitch_is_tensor else float(pitch_control)
            energy_scale = 1.0 if energy_control is None or energy_is_tensor else float(energy_control)
            dur_pred = None if dur_is_tensor else torch.empty(B, 1, P, **f32)
            f0 = torch.empty(B, P, **f32)
            energy = torch.empty(B, P, **f32)
            pitch_in = energy_in = None
            if pitch_is_tensor:
                pitch_in = pitch_control.to(dev, torch.float32).contiguous()
                if tuple(pitch_in.shape) != (B, P):
                    raise ValueError("pitch_control tensor must be [B, P] (Hz per phoneme)")
            if energy_is_tensor:
                energy_in = energy_control.to(dev, torch.float32).contiguous()
                if tuple(energy_in.shape) != (B, P):
                    raise ValueError("energy_control tensor must be [B, P]")
            N.check(lib.vits_variance_adaptor(
                h, x.data_ptr(), lens.data_ptr(), cond.data_ptr(), B, P,
                1 if dur_is_tensor else 0, dur_scale, N.ptr(dur_pred),
                1 if pitch_is_tensor else 0, pitch_scale, N.ptr(pitch_in), f0.data_ptr(),
                1 if energy_is_tensor else 0, energy_scale, N.ptr(energy_in), energy.data_ptr(),
                ws.data_ptr(), ws.numel(), stream), "variance_adaptor")
            if stages is not None:
                stages["x_phone"] = x.clone()
            # length regulator: plan on device, one small D2H copy to size the frame tensors
            if dur_is_tensor:
                duration = duration_control
                d = duration_control
                if d.dtype not in _DUR_DTYPES:
                    d = d.to(torch.float32 if d.is_floating_point() else torch.int64)
                d = d.to(dev).contiguous()
                if d.numel() != B * P:
                    raise ValueError("duration_control tensor must hold one value per phoneme ([B, P] or [B, 1, P])")
            else:
                duration = dur_pred
                d = dur_pred
            cum = torch.empty(B, P, dtype=torch.int32, device=dev)
            totals = torch.empty(B, dtype=torch.int32, device=dev)
            report = torch.empty(B + 2, dtype=torch.int32, device=dev)
            N.check(lib.vits_length_regulate_plan_report(d.data_ptr(), _DUR_DTYPES[d.dtype], B, P, cum.data_ptr(),
                                                         totals.data_ptr(), report.data_ptr(), err.data_ptr(),
                                                         self._range.data_ptr(), stream), "length_regulate_plan")
            # one device-to-host copy: frame counts, id-range errors, and the fp16-range flag that generator calls
            # before this one may have raised (synchronises the stream); the plan kernel has put the three side by side
            host = self._read_back(report)
            self._err_pending = False
            if int(host[-2]):
                err.zero_()
            if int(host[-2]) & 1:
                raise IndexError("phoneme id out of range of the symbol table")
            if int(host[-2]) & 2:
                raise IndexError("speaker id out of range")
            self._report_range(int(host[-1]))
            T = int(host[:-2].max())
            if T <= 0:
                raise ValueError("all durations are zero: nothing to synthesise")
            x_frame = torch.empty(B, Cc, T, **f32)
            N.check(lib.vits_length_regulate(x.data_ptr(), cum.data_ptr(), totals.data_ptr(), B, Cc, P, T,
                                             x_frame.data_ptr(), stream), "length_regulate")
            if stages is not None:
                stages["x_frame"] = x_frame.clone()
                stages["x_lengths"] = totals.to(torch.int64)
            # frame prior, projection, sampling
            if noise is None:
                noise_t = torch.randn(B, IC, T, **f32)
            else:
                noise_t = noise.to(dev, torch.float32).contiguous()
                if tuple(noise_t.shape) != (B, IC, T):
                    raise ValueError(f"noise must be [{B}, {IC}, {T}]")
            t_dec = len(range(T)[:max_len])
            if t_dec <= 0:
                raise ValueError("max_len leaves no frames to decode")
            need = max(lib.vits_encoder_workspace_bytes(h, B, T), lib.vits_flow_workspace_bytes(h, B, T))
            ws = self._workspace(need)
            stats = torch.empty(B, 2 * IC, T, **f32)
            z_p = torch.empty(B, IC, T, **f32)
            z = torch.empty(B, IC, T, **f32)
            N.check(lib.vits_frame_prior(h, x_frame.data_ptr(), totals.data_ptr(), B, T, noise_t.data_ptr(),
                                         float(noise_scale), stats.data_ptr(), z_p.data_ptr(), z.data_ptr(),
                                         ws.data_ptr(), ws.numel(), stream), "frame_prior")
            if stages is not None:
                stages["prior"] = x_frame.clone()
            N.check(lib.vits_flow_reverse(h, z.data_ptr(), totals.data_ptr(), cond.data_ptr(), B, T, ws.data_ptr(),
                                          ws.numel(), stream), "flow_reverse")
            x_mask = (torch.arange(T, device=dev)[None, :] < totals[:, None]).unsqueeze(1)
            m_p, logs_p = stats[:, :IC], stats[:, IC:]
        return dict(B=B, T=T, t_dec=t_dec, z=z, totals=totals, cond=cond,
                    outputs=(x_mask, (z, z_p, m_p, logs_p), duration, f0, energy))


def config_from_hparams(hps, n_vocab: int):
    """ModelConfig from a reference config (utils.get_hparams_from_file) the way the reference's callers build the
    model: SynthesizerTrn(len(symbols), filter_length // 2 + 1, hop_length, sampling_rate, segment_size // hop_length,
    n_speakers=..., **hps.model) (reference inference.py:25-33)."""
    model = dict(hps.model.items()) if hasattr(hps.model, "items") else dict(hps.model)
    fields = {f.name for f in __import__("dataclasses").fields(synthetic.ModelConfig)}
    kw = {k: v for k, v in model.items() if k in fields}
    for k in ("resblock_kernel_sizes", "upsample_rates", "upsample_kernel_sizes"):
        if k in kw:
            kw[k] = tuple(kw[k])
    if "resblock_dilation_sizes" in kw:
        kw["resblock_dilation_sizes"] = tuple(tuple(d) for d in kw["resblock_dilation_sizes"])
    if "resblock" in kw:
        kw["resblock"] = str(kw["resblock"])
    return synthetic.ModelConfig(n_vocab=n_vocab, spec_channels=hps.data.filter_length // 2 + 1,
                                 hop_length=hps.data.hop_length, sampling_rate=hps.data.sampling_rate,
                                 segment_size=hps.train.segment_size // hps.data.hop_length,
                                 n_speakers=hps.data.n_speakers, **kw)


# Frames of context the generator needs on each side of a window for its output to equal that of the whole
# utterance.  Walking back from the waveform through the reference configuration (reference models.py:257-303,
# modules.py:183-225): conv_post 3 samples; each ResBlock stage 60 (kernel 11: (5+5) + (15+5) + (25+5)); a transposed
# conv of stride s turns a reach of R output samples into ceil(R / s) + 1 input samples; conv_pre 3 frames:
# 63 -> 33 -> 93 -> 25 -> 85 -> 12 -> 72 -> 10 -> 13 frames.  InferStream computes it from the model's own configuration.
def generator_halo_frames(cfg) -> int:
    reach = 3                                                     # conv_post, kernel 7
    second = 0 if str(cfg.resblock) == "2" else 1                 # ResBlock2 has no second, undilated conv per dilation
    per_stage = max(sum((k - 1) * d // 2 + second * ((k - 1) // 2) for d in dils)
                    for k, dils in zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes))
    for u in reversed(cfg.upsample_rates):
        reach = -(-(reach + per_stage) // u) + 1
    return reach + 3                                              # conv_pre, kernel 7


class InferStream:
    """Iterator over the waveform windows of one SynthesizerTrn.infer_stream call (see there)."""

    def __init__(self, net: "SynthesizerTrn", fe: dict, chunk_frames: int):
        self._net, self._fe, self.chunk_frames = net, fe, chunk_frames
        self.halo_frames = generator_halo_frames(net.cfg)
        self.outputs = fe["outputs"]
        self.total_frames = fe["t_dec"]
        self.total_samples = fe["t_dec"] * net.total_upsample

    def __iter__(self):
        net, fe = self._net, self._fe
        lib, h, dev, up = N.lib, net._handle, net.device, net.total_upsample
        B, T, t_dec, z, H = fe["B"], fe["T"], fe["t_dec"], fe["z"], self.halo_frames
        with torch.no_grad(), torch.cuda.device(dev):
            for s in range(0, t_dec, self.chunk_frames):
                e = min(s + self.chunk_frames, t_dec)
                s0, e0 = max(0, s - H), min(t_dec, e + H)          # window + context; the utterance's own ends stay
                n = e0 - s0
                with net._exclusive():                             # the lock is free again between windows
                    net._budget_generator_streams(torch.cuda.current_stream(dev))
                    stream = torch.cuda.current_stream(dev).cuda_stream
                    ws = net._workspace(lib.vits_generator_workspace_bytes(h, B, n))
                    lens = (fe["totals"] - s0).clamp_(min=0)       # the length mask, relative to the window start
                    o = torch.empty(B, 1, n * up, dtype=torch.float32, device=dev)
                    # window of z: same row stride T, start shifted by s0 frames
                    N.check(lib.vits_generator(h, z.data_ptr() + 4 * s0, lens.data_ptr(), fe["cond"].data_ptr(), B, T,
                                               n, o.data_ptr(), ws.data_ptr(), ws.numel(), stream),
                            "generator (window)")
                yield s * up, o[:, :, (s - s0) * up:(e - s0) * up]
