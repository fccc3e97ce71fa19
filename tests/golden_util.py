"""Helpers for reading the committed golden fixtures (tests/golden/*.npz, written by tools/make_goldens.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_CASES = ["small_b1", "ragged_b2", "predicted_scalar", "tensor_controls", "max_len", "tiny"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta_json"}
    meta = json.loads(bytes(z["meta_json"]).decode()) if "meta_json" in z.files else {}
    return d, meta


def infer_kwargs(d, meta):
    """Keyword arguments of the infer call the fixture was made with."""
    kw = dict(meta)
    for k in ("duration_control", "pitch_control", "energy_control"):
        if "in_" + k in d:
            kw[k] = d["in_" + k]
    return kw
