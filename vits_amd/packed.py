"""One-file packed weights: the result of the load-time preparation (weight-norm folded, Flip / q-k-v / res-skip
reordering done, every conv in MFMA fragment order for its arithmetic mode) written once and mapped back without
touching the checkpoint again.

The reference reloads its training checkpoint -- ``torch.save({'model': state_dict, 'iteration', 'optimizer',
'learning_rate'})``, reference utils.py:52-68 -- and re-evaluates weight-norm on every forward.  Converting once:

    python -m vits_amd.packed G_76000.pth configs/config.json model.vitsw [--precision f16x2]

and then ``SynthesizerTrn.from_packed("model.vitsw")`` uploads the mapped bytes as they are.

File layout: 8-byte magic ``VITSAMD1``, uint64 little-endian header length, a JSON header, then the tensors, each at a
256-byte-aligned offset counted from the end of the header.  The header holds the format version, the model
configuration, the model-level precision and one record per tensor: ``{name, dtype, numel, offset}`` for plain
tensors, for convs ``{name, dtype, numel, offset, bias_offset (or -1), cin, rows, taps, dilation, mode,
precision}``, and ``{kind: "alt", name, dtype, numel, offset}`` for the second (f16x2) packing some front-end convs
carry."""
from __future__ import annotations

import dataclasses
import json
import mmap
import os
import struct
import sys
from typing import Dict, Tuple

import numpy as np
import torch

from . import _native as N
from . import synthetic, weights

MAGIC = b"VITSAMD1"
VERSION = 1
_ALIGN = 256
_DTYPES = {"float32": (torch.float32, np.float32, 4), "int16": (torch.int16, np.int16, 2)}


def _cfg_to_json(cfg) -> dict:
    d = dataclasses.asdict(cfg)
    return {k: (list(map(list, v)) if k == "resblock_dilation_sizes" else list(v) if isinstance(v, tuple) else v)
            for k, v in d.items()}


def _cfg_from_json(d: dict):
    fields = {f.name for f in dataclasses.fields(synthetic.ModelConfig)}
    unknown = set(d) - fields
    if unknown:
        raise ValueError(f"packed weights: unknown configuration keys {sorted(unknown)}")
    kw = {}
    for k, v in d.items():
        if k == "resblock_dilation_sizes":
            v = tuple(tuple(x) for x in v)
        elif isinstance(v, list):
            v = tuple(v)
        kw[k] = v
    return synthetic.ModelConfig(**kw)


def save_packed(path: str, state_dict: Dict[str, torch.Tensor], cfg=None, precision: str = "f16x2") -> dict:
    """Prepare `state_dict` (reference checkpoint layout) for `cfg` and write the packed file.  Every tensor the
    synthesis path reads must be present with the shape the configuration implies.  Returns the header."""
    cfg = cfg or synthetic.ModelConfig()
    if precision not in N.PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(N.PRECISIONS)}")
    manifest = synthetic.state_dict_manifest(cfg)
    for k, shape in manifest.items():
        if k.startswith(weights.UNUSED_PREFIXES):
            continue
        if k not in state_dict:
            raise KeyError(f"checkpoint is missing {k}")
        if tuple(state_dict[k].shape) != tuple(shape):
            raise ValueError(f"size mismatch for {k}: checkpoint {tuple(state_dict[k].shape)}, model {tuple(shape)}")
    pw = weights.prepare(state_dict, cfg, N.PRECISIONS[precision])
    records, blobs, off = [], [], 0

    def place(t: torch.Tensor) -> int:
        nonlocal off
        start = off
        raw = t.contiguous().numpy().tobytes()
        blobs.append((start, raw))
        off = (start + len(raw) + _ALIGN - 1) // _ALIGN * _ALIGN
        return start

    for name, t in pw.plain.items():
        records.append(dict(kind="plain", name=name, dtype="float32", numel=t.numel(), offset=place(t)))
    for name, (w, b, cin, rows, taps, dil, mode, prec) in pw.convs.items():
        dt = "float32" if w.dtype == torch.float32 else "int16"
        rec = dict(kind="conv", name=name, dtype=dt, numel=w.numel(), offset=place(w), bias_offset=-1, cin=cin, rows=rows,
                   taps=taps, dilation=dil, mode=mode, precision=prec)
        if b is not None:
            rec["bias_offset"] = place(b)
        records.append(rec)
    for name, w in pw.alts.items():           # second (f16x2) packing of a conv recorded above
        records.append(dict(kind="alt", name=name, dtype="int16", numel=w.numel(), offset=place(w)))
    header = dict(version=VERSION, precision=precision, config=_cfg_to_json(cfg), data_bytes=off, tensors=records)
    hb = json.dumps(header).encode()
    hb += b" " * (-(16 + len(hb)) % _ALIGN)                   # data starts on a 256-byte boundary of the file
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(hb)))
        f.write(hb)
        base = f.tell()
        for start, raw in blobs:
            f.seek(base + start)
            f.write(raw)
        f.truncate(base + off)
    os.replace(tmp, path)
    return header


def load_packed(path: str) -> Tuple[object, str, weights.PreparedWeights]:
    """Map a packed file: returns (ModelConfig, precision name, PreparedWeights whose tensors are views of the
    mapping -- nothing is copied until the upload)."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) < 16 or head[:8] != MAGIC:
            raise ValueError(f"{path}: not a vits_amd packed weight file")
        hlen = struct.unpack("<Q", head[8:])[0]
        if hlen <= 0 or 16 + hlen > size:
            raise ValueError(f"{path}: truncated header")
        header = json.loads(f.read(hlen).decode())
        if header.get("version") != VERSION:
            raise ValueError(f"{path}: format version {header.get('version')} (this build reads {VERSION})")
        base = 16 + hlen
        if base + header["data_bytes"] > size:
            raise ValueError(f"{path}: truncated ({size} bytes, header promises {base + header['data_bytes']})")
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    cfg = _cfg_from_json(header["config"])
    precision = header["precision"]
    if precision not in N.PRECISIONS:
        raise ValueError(f"{path}: unknown precision {precision!r}")

    def view(offset: int, numel: int, dtype: str) -> torch.Tensor:
        tdt, ndt, isz = _DTYPES[dtype]
        if offset < 0 or base + offset + numel * isz > size:
            raise ValueError(f"{path}: tensor outside the file")
        arr = np.frombuffer(mm, dtype=ndt, count=numel, offset=base + offset)
        return torch.from_numpy(arr)        # read-only view; only ever copied to the device

    pw = weights.PreparedWeights(N.PRECISIONS[precision])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # torch warns about non-writable numpy views; they are never written
        for r in header["tensors"]:
            t = view(r["offset"], r["numel"], r["dtype"])
            if r["kind"] == "plain":
                pw.plain[r["name"]] = t
            elif r["kind"] == "alt":
                pw.alts[r["name"]] = t
            else:
                b = None if r["bias_offset"] < 0 else view(r["bias_offset"], r["rows"], "float32")
                pw.convs[r["name"]] = (t, b, r["cin"], r["rows"], r["taps"], r["dilation"], r["mode"], r["precision"])
    pw._mapping = mm                         # keeps the mapping alive as long as the views
    return cfg, precision, pw


def main(argv=None):
    import argparse
    from . import utils
    ap = argparse.ArgumentParser(description="fold and pack a reference checkpoint into one mappable file")
    ap.add_argument("checkpoint", help="G_*.pth written by the reference's utils.save_checkpoint")
    ap.add_argument("config", help="the run's config.json")
    ap.add_argument("out")
    ap.add_argument("--precision", default="f16x2", choices=sorted(N.PRECISIONS))
    a = ap.parse_args(argv)
    hps = utils.get_hparams_from_file(a.config)
    from .models import config_from_hparams
    ck = torch.load(a.checkpoint, map_location="cpu", weights_only=True)
    sd = ck["model"] if "model" in ck else ck
    cfg = config_from_hparams(hps, n_vocab=sd["enc_p.symbol_emb.weight"].shape[0])     # len(symbols) in the reference
    h = save_packed(a.out, sd, cfg, a.precision)
    print(f"{a.out}: {len(h['tensors'])} tensors, {h['data_bytes'] / 1e6:.1f} MB, precision {a.precision}")


if __name__ == "__main__":
    main(sys.argv[1:])
