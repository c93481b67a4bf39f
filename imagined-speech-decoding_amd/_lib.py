"""ctypes binding of libisd_hip.so (the C ABI of include/isd_hip.h).

There is no CPU fallback: if the library is missing or a call fails the
product raises.  Build it with ``python -m isd_amd.build`` (or
``__graft_entry__.build()``).
"""
import ctypes as C
import os

from .build import LIB_PATH

ISD_OK = 0
ISD_ERR_INVALID, ISD_ERR_UNSUPPORTED, ISD_ERR_HIP, ISD_ERR_NO_DEVICE = -1, -2, -3, -4
FB_F32, FB_F64, FB_AUTO, FB_MIXED = 0, 1, 2, 3
BP_MAGNITUDE, BP_POWER, BP_LOGPOWER = 0, 1, 2


class IsdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libisd_hip error {code}: {msg}")
        self.code = code


class IsdUnsupported(IsdError):
    pass


_p, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_pi, _pd = C.POINTER(C.c_int), C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/isd_hip.h declares
SIGNATURES = {
    "isd_abi_version": (_i, []),
    "isd_last_error": (C.c_char_p, []),
    "isd_shader_clock_probe": (_i, [_p, _i, _p]),
    "isd_wall_clock_khz": (_i, []),
    "isd_exact_sum_workspace_bytes": (_i64, []),
    "isd_exact_sum": (_i, [_p, _i64, _p, _p, _p]),
    "isd_device_count": (_i, []),
    "isd_adamw_step": (_i, [_p, _p, _p, _p, _i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i64,
                       _p, _p, _p]),
    "isd_adamw_multi_step": (_i, [_i, _p, _p, _p, _p, _p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                             _i64, _p, _p, _p]),
    "isd_fb_plan_create": (_i, [C.POINTER(_p), _i, _i, _pd, _pd, _i]),
    "isd_fb_plan_destroy": (_i, [_p]),
    "isd_fb_plan_precision": (_i, [_p]),
    "isd_fb_forward": (_i, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "isd_stft_plan_create": (_i, [C.POINTER(_p), _i, _i, _i]),
    "isd_stft_plan_destroy": (_i, [_p]),
    "isd_stft_plan_frames": (_i, [_p]),
    "isd_stft_plan_bins": (_i, [_p]),
    "isd_stft_forward": (_i, [_p, _p, _p, _i64, _p]),
    "isd_stft_bandpower": (_i, [_p, _p, _p, _i64, _i64, _i, _i, _pi, _pi, _i, _f, _p]),
    "isd_features_fused": (_i, [_p, _p, _p, _p, _i64, _i64, _pi, _pi, _i, _f, _p]),
    "isd_features_fused_bf16": (_i, [_p, _p, _p, _p, _i64, _i64, _pi, _pi, _i, _f, _p]),
    "isd_features_fused_last_path": (_i, []),
    "isd_fir_plan_create": (_i, [C.POINTER(_p), _i, _pd]),
    "isd_fir_plan_destroy": (_i, [_p]),
    "isd_fir_plan_taps": (_i, [_p]),
    "isd_fir_zero_phase_f32": (_i, [_p, _p, _p, _i64, _i, _p]),
    "isd_fir_zero_phase_f64": (_i, [_p, _p, _p, _i64, _i, _p]),
    "isd_conv4_plan_create": (_i, [C.POINTER(_p), _i, _i, _pi, _pi, _i, _i, _i, _i]),
    "isd_conv4_plan_destroy": (_i, [_p]),
    "isd_conv4_plan_set_activation_dtype": (_i, [_p, _i]),
    "isd_conv4_param_count": (_i64, [_p]),
    "isd_conv4_param_offset": (_i64, [_p, _i, _i]),
    "isd_conv4_windows": (_i, [_p, _i64]),
    "isd_conv4_workspace_bytes": (_i64, [_p, _i64, _i64]),
    "isd_conv4_forward": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _p]),
    "isd_conv4_backward": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    "isd_conv4_backward_x": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    "isd_linear_forward": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _p]),
    "isd_linear_residual_forward": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _p]),
    "isd_linear_workspace_bytes": (_i64, [_i64, _i, _i]),
    "isd_embed_forward": (_i, [_p, _p, _p, _p, _i64, _i, _i, _p]),
    "isd_embed_backward": (_i, [_p, _p, _p, _p, _i64, _i, _i, _p]),
    "isd_layernorm_forward": (_i, [_p, _p, _p, _p, _p, _i64, _i, _f, _p]),
    "isd_layernorm_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _p]),
    "isd_attention_forward": (_i, [_p, _p, _p, _i64, _i, _i, _i, _f, C.c_uint64, _p]),
    "isd_attention_backward": (_i, [_p, _p, _p, _p, _i64, _i, _i, _i, _f, C.c_uint64, _p]),
    "isd_tail_fused_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "isd_tail_fused_param_count": (_i64, [_i, _i, _i, _i]),
    "isd_tail_fused_save_floats": (_i64, [_i64, _i, _i, _i]),
    "isd_tail_fused_workspace_floats": (_i64, [_i64, _i, _i, _i, _i, _i]),
    "isd_tail_fused_forward": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, C.c_uint64, _p,
                                    _p]),
    "isd_tail_fused_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f,
                                     C.c_uint64, _p, _p]),
    "isd_linear_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _p]),
    "isd_eegnet_plan_create": (_i, [C.POINTER(_p), _i, _i, _i, _i]),
    "isd_featcnn_supported": (_i, [_p, _i64, _i64, _i]),
    "isd_featcnn_step": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _f, _p]),
    "isd_featcnn_step_bf16": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _f, _p]),
    "isd_paperhead_plan_create": (_i, [C.POINTER(_p), _i, _i, _i]),
    "isd_paperhead_plan_destroy": (_i, [_p]),
    "isd_paperhead_param_count": (_i64, [_p]),
    "isd_paperhead_buffer_count": (_i64, [_p]),
    "isd_paperhead_workspace_bytes": (_i64, [_p, _i64]),
    "isd_paperhead_forward": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _f, _f, _p]),
    "isd_paperhead_backward": (_i, [_p, _p, _p, _p, _p, _p, _i64, _p]),
    "isd_paperhead_backward_x": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _p]),
    "isd_paperhead_forward_stage": (_i, [_p, _i, _p, _p, _p, _p, _p, _i64, _i, _f, _f, _i, _p]),
    "isd_paperhead_backward_stage": (_i, [_p, _i, _p, _p, _p, _p, _p, _i64, _i, _p]),
    "isd_paperhead_sync_block": (_i, [_p, _i64, _i, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "isd_paperhead_sync_block_kind": (_i, [_i, _i]),
    "isd_eegnet_sync_block_kind": (_i, [_i, _i]),
    "isd_zone_batch_begin": (_i, []),
    "isd_zone_batch_next": (_i, []),
    "isd_zone_batch_launch": (_i, [_p]),
    "isd_zone_batch_abort": (_i, []),
    "isd_cvblock_plan_create": (_i, [C.POINTER(_p), _i, _i, _i]),
    "isd_cvblock_flat_dim": (_i64, [_p]),
    "isd_eegnet_plan_set_seed_counter": (_i, [_p, _p]),
    "isd_eegnet_plan_destroy": (_i, [_p]),
    "isd_eegnet_param_count": (_i64, [_p]),
    "isd_eegnet_buffer_count": (_i64, [_p]),
    "isd_eegnet_workspace_bytes": (_i64, [_p, _i64]),
    "isd_eegnet_forward": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _f, _f, _f, C.c_uint64, _p]),
    "isd_eegnet_backward": (_i, [_p, _p, _p, _p, _p, _p, _i64, _f, C.c_uint64, _p]),
    "isd_eegnet_backward_x": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _f, C.c_uint64, _p]),
    "isd_eegnet_forward_stage": (_i, [_p, _i, _p, _p, _p, _p, _p, _i64, _i, _f, _f, _f, C.c_uint64, _i, _p]),
    "isd_eegnet_backward_stage": (_i, [_p, _i, _p, _p, _p, _p, _p, _i64, _f, C.c_uint64, _i, _p]),
    "isd_eegnet_sync_block": (_i, [_p, _i64, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "isd_softmax_ce_workspace_bytes": (_i64, [_i64]),
    "isd_softmax_ce": (_i, [_p, _p, _i, _p, _p, _p, _p, _i64, _i, _i, _f, _p, _p]),
}

_lib = None


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU fallback for the product path.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)           # AttributeError if the header and the .so disagree
            fn.restype, fn.argtypes = res, args
        if h.isd_abi_version() != 1:
            raise ImportError(f"{LIB_PATH}: ABI version {h.isd_abi_version()} != 1; rebuild")
        _lib = h
    return _lib


def check(rc):
    if rc != ISD_OK:
        msg = lib().isd_last_error().decode("utf-8", "replace")
        raise (IsdUnsupported if rc == ISD_ERR_UNSUPPORTED else IsdError)(rc, msg)
    return rc


def int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def double_array(vals):
    return (C.c_double * len(vals))(*[float(v) for v in vals])
