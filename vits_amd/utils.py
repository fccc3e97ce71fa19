"""Config and checkpoint I/O with the reference's interface (reference utils.py:19-49, 217-223, 280-309)."""
from __future__ import annotations

import json
import logging
import os
import re

import torch

logger = logging.getLogger("vits_amd")


class HParams:
    """Nested attribute / item access over the config dict (same behaviour as reference utils.py:280-309)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            self[k] = HParams(**v) if isinstance(v, dict) else v

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def __len__(self):
        return len(self.__dict__)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        return setattr(self, key, value)

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        return self.__dict__.__repr__()


def _read_config_text(config_path: str) -> str:
    """The reference's shipped configs/config.json is GBK-encoded and carries // and /* */ comments, which
    plain open().read() + json.loads (reference utils.py:217-223) rejects; accept it as well as clean UTF-8 JSON."""
    raw = open(config_path, "rb").read()
    for enc in ("utf-8", "gbk"):
        try:
            text = raw.decode(enc)
            break
        except UnicodeDecodeError:
            continue
    else:
        raise UnicodeDecodeError("utf-8/gbk", raw, 0, 1, f"cannot decode {config_path}")
    try:
        json.loads(text)
        return text
    except json.JSONDecodeError:
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        return re.sub(r"(?m)//[^\n\"]*$", "", text)


def get_hparams_from_file(config_path: str) -> HParams:
    return HParams(**json.loads(_read_config_text(config_path)))


def load_checkpoint(checkpoint_path, model, optimizer=None, skip_optimizer=False):
    """Same contract as reference utils.py:19-49: read {'model', 'iteration', 'learning_rate', ...}, copy every key
    the model knows from the file, keep (and report) the model's own value for keys that are absent or mis-shaped.

    The file is read with torch.load(weights_only=True): nothing in it is executed."""
    assert os.path.isfile(checkpoint_path)
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    if "model" in ckpt:
        saved, iteration, lr = ckpt["model"], ckpt.get("iteration", 0), ckpt.get("learning_rate", 0.0)
    else:                      # a bare state_dict
        saved, iteration, lr = ckpt, 0, 0.0
    target = model.module if hasattr(model, "module") else model
    own = target.state_dict()
    new_sd = {}
    for k, v in own.items():
        if k in saved and (v is None or tuple(saved[k].shape) == tuple(v.shape)):
            new_sd[k] = saved[k]
        else:
            print("error, %s is not in the checkpoint" % k)
            logger.info("%s is not in the checkpoint" % k)
            if v is not None:
                new_sd[k] = v
    target.load_state_dict(new_sd)
    logger.info("Loaded checkpoint '%s' (iteration %s)", checkpoint_path, iteration)
    return model, optimizer, lr, iteration
