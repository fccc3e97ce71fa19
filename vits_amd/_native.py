"""ctypes binding of libvits_amd.so (C ABI in include/vits_amd.h).

There is no fallback: if the shared library is missing or fails to load, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported before the library is opened: both link libamdhip64.so.7, and the process must end up with
# ONE HIP runtime -- the one PyTorch ships -- or device pointers and streams could not be shared between the two.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITS_AMD_LIB") or os.path.join(_HERE, "lib", "libvits_amd.so")


class VitsConfig(C.Structure):
    _fields_ = [
        ("n_vocab", C.c_int), ("hidden_channels", C.c_int), ("inter_channels", C.c_int),
        ("filter_channels", C.c_int), ("n_heads", C.c_int), ("n_layers", C.c_int), ("kernel_size", C.c_int),
        ("gin_channels", C.c_int), ("n_speakers", C.c_int),
        ("pitch_layers", C.c_int), ("dur_filter", C.c_int), ("energy_filter", C.c_int),
        ("n_flows", C.c_int), ("flow_layers", C.c_int), ("flow_kernel", C.c_int), ("window_size", C.c_int),
        ("n_upsamples", C.c_int),
        ("upsample_rates", C.c_int * 8), ("upsample_kernel_sizes", C.c_int * 8),
        ("upsample_initial_channel", C.c_int),
        ("n_resblock_kernels", C.c_int), ("resblock_kernel_sizes", C.c_int * 8),
        ("n_resblock_dilations", C.c_int), ("resblock_dilation_sizes", (C.c_int * 8) * 8),
        ("resblock", C.c_int),
    ]


class VitsError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m vits_amd.build` (hipcc, gfx950). "
            "vits_amd has no CPU or PyTorch fallback path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    sig = {
        "vits_last_error": (C.c_char_p, []),
        "vits_version": (C.c_char_p, []),
        "vits_model_create": (i32, [C.POINTER(VitsConfig), C.POINTER(vp)]),
        "vits_model_destroy": (None, [vp]),
        "vits_model_alive": (i32, [vp]),
        "vits_model_set_generator_streams": (i32, [vp, i32]),
        "vits_model_set_front_end_overlapped": (i32, [vp, i32]),
        "vits_model_set_ln_fused": (i32, [vp, i32]),
        "vits_model_set_range_flag": (i32, [vp, vp]),
        "vits_model_set_tensor": (i32, [vp, C.c_char_p, vp, i64]),
        "vits_model_set_conv": (i32, [vp, C.c_char_p, vp, vp, i32, i32, i32, i32, i32, i32]),
        "vits_model_set_conv_alt": (i32, [vp, C.c_char_p, vp]),
        "vits_set_front_end_f16x2": (None, [i32]),
        "vits_model_finalize": (i32, [vp]),
        "vits_pack_conv_weight": (i32, [vp, i32, i32, i32, vp]),
        "vits_pack_conv_weight_bf16x3": (i32, [vp, i32, i32, i32, vp]),
        "vits_pack_conv_weight_bf16": (i32, [vp, i32, i32, i32, vp]),
        "vits_pack_conv_weight_f16x2": (i32, [vp, i32, i32, i32, vp]),
        "vits_cond_rows": (i32, [vp]),
        "vits_speaker_cond": (i32, [vp, vp, i32, vp, vp, vp]),
        "vits_encoder_workspace_bytes": (i64, [vp, i32, i32]),
        "vits_text_encoder": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, i64, vp]),
        "vits_encoder_stack": (i32, [vp, i32, vp, vp, i32, i32, vp, i64, vp]),
        "vits_variance_workspace_bytes": (i64, [vp, i32, i32]),
        "vits_variance_adaptor": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp, i32, f32, vp, vp, i32, f32, vp, vp,
                                        vp, i64, vp]),
        "vits_length_regulate_plan": (i32, [vp, i32, i32, i32, vp, vp, vp]),
        "vits_length_regulate_plan_report": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
        "vits_length_regulate": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
        "vits_frame_prior": (i32, [vp, vp, vp, i32, i32, vp, f32, vp, vp, vp, vp, i64, vp]),
        "vits_flow_workspace_bytes": (i64, [vp, i32, i32]),
        "vits_flow_reverse": (i32, [vp, vp, vp, vp, i32, i32, vp, i64, vp]),
        "vits_flow_forward": (i32, [vp, vp, vp, vp, i32, i32, vp, i64, vp]),
        "vits_flow_noalloc_max_positions": (i64, []),
        "vits_flow_reverse_noalloc": (i32, [vp, vp, vp, vp, i32, i32, vp]),
        "vits_posterior_channels": (i32, [vp]),
        "vits_posterior_workspace_bytes": (i64, [vp, i32, i32]),
        "vits_posterior_encoder": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp]),
        "vits_generator_workspace_bytes": (i64, [vp, i32, i32]),
        "vits_generator": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, i64, vp]),
        "vits_profile_enable": (None, [i32]),
        "vits_profile_reset": (None, []),
        "vits_profile_read_bytes": (i32, [i32, C.POINTER(C.c_double)]),
        "vits_profile_read": (i32, [i32, C.c_char_p, i32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int64)]),
        "vits_op_set_range_flag": (None, [vp]),
        "vits_op_conv1d": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, i32, f32, f32, vp]),
        "vits_op_conv1d_out2": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, i32, f32, f32, f32, vp]),
        "vits_op_resblock_pair": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, i32, i32, C.c_float, vp]),
        "vits_op_resblock_pair_prec": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, C.c_float, vp]),
        "vits_op_resblock_chain": (i32, [vp, vp, vp, vp, i32, i32, vp, i32, i32, vp, vp, i32, i32, i32, C.c_float, vp]),
        "vits_op_layer_norm": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "vits_op_attention": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "vits_op_attention_prec": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)      # AttributeError here means the library does not match include/vits_amd.h
        fn.restype = res
        fn.argtypes = args
    return lib, list(sig.keys())


lib, EXPORTED = _load()

MODE_PLAIN, MODE_GATE, MODE_UP8, MODE_UP4, MODE_UP2 = 0, 1, 2, 3, 4
PREC_F32, PREC_BF16X3, PREC_BF16, PREC_F16X2 = 0, 1, 2, 3
PRECISIONS = {"fp32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16, "f16x2": PREC_F16X2}
F_IN_LRELU, F_IN_MASK, F_RELU, F_MASK_PRE, F_RES, F_RES_SUB, F_ACCUM, F_MASK_POST = (1 << i for i in range(8))
F_OUT_STAGED, F_IN_STAGED = 1 << 8, 1 << 9


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise VitsError(f"{what or 'vits_amd'} failed ({rc}): {lib.vits_last_error().decode()}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor / None."""
    return 0 if t is None else t.data_ptr()


def profile_rows():
    """Resolve and return the per-kernel timing rows: list of dict(name, flops, ms, count, bytes)."""
    n = lib.vits_profile_read(-1, None, 0, None, None, None)
    rows = []
    for i in range(n):
        name = C.create_string_buffer(128)
        fl, ms, cnt = C.c_double(), C.c_double(), C.c_int64()
        check(lib.vits_profile_read(i, name, 128, C.byref(fl), C.byref(ms), C.byref(cnt)), "profile_read")
        by = C.c_double()
        check(lib.vits_profile_read_bytes(i, C.byref(by)), "profile_read_bytes")
        rows.append(dict(name=name.value.decode(), flops=fl.value, ms=ms.value, count=cnt.value, bytes=by.value))
    return rows
