"""MI355X-native implementation of the MVTS (xirui123/VITS) synthesis path.

    from vits_amd import SynthesizerTrn, utils      # same surface as the reference's models / utils modules

`SynthesizerTrn.infer` runs entirely in libvits_amd.so (hand-written gfx950 HIP kernels behind the C ABI in
include/vits_amd.h).  Build the library in-tree with `python -m vits_amd.build`.
"""
__all__ = ["SynthesizerTrn", "utils", "synthetic"]


def __getattr__(name):
    if name == "SynthesizerTrn":
        from .models import SynthesizerTrn
        return SynthesizerTrn
    if name in ("utils", "synthetic", "models", "weights"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
