"""ctypes binding of ``libbuddy_hip.so`` (C-ABI in ``include/buddy_hip.h``).

There is NO CPU fallback: importing works anywhere (so configs/host logic can be tested on CPU), but any
compute call raises ``BuddyHipError`` if the shared library is missing or no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbuddy_hip.so")


class BuddyHipError(RuntimeError):
    pass


_lib = None

_f32p = C.c_void_p  # raw device / host pointers are passed as integers
_SIGS = {
    "buddy_last_error": (C.c_char_p, []),
    "buddy_version": (C.c_int, []),
    "buddy_ncsnpp_param_count": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    "buddy_ncsnpp_create": (C.c_int, [_f32p, C.c_longlong, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]),
    "buddy_ncsnpp_destroy": (C.c_int, [C.c_void_p]),
    "buddy_ncsnpp_replica": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "buddy_ncsnpp_weight_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                            C.POINTER(C.c_int)]),
    "buddy_conv3_weight_prep": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_void_p]),
    "buddy_ncsnpp_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    "buddy_ncsnpp_forward": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_ncsnpp_vjp": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_void_p]),
    "buddy_ncsnpp_tap": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int * 4)]),
    "buddy_prof_enable": (C.c_int, [C.c_int]),
    "buddy_wpe": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_wpe_workspace_bytes": (C.c_longlong, [C.c_int, C.c_int]),
    "buddy_wpe_dereverb": (C.c_int, [_f32p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_prof_collect_wino4": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_longlong)]),
    "buddy_prof_collect_hbm": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "buddy_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "buddy_copy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "buddy_hbm_ubench": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_mfma_ubench": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "buddy_mfma_ubench_bf16": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "buddy_gemm": (C.c_int, [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                             _f32p, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_void_p]),
    "buddy_gemm_winograd_domain": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_wgemm_packed_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "buddy_wgemm_pack_weights": (C.c_int, [_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_gemm_winograd_domain_bf16x3": (C.c_int, [_f32p, C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_wgemm_f16x2_packed_bytes": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "buddy_wgemm_f16x2_pack_weights": (C.c_int, [_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_gemm_winograd_domain_f16x2": (C.c_int, [_f32p, C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "buddy_abs_max_bits": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]),
    "buddy_gemm_bf16x3": (C.c_int, [_f32p, C.c_int, _f32p, C.c_int, C.c_int, C.c_void_p, _f32p, C.c_int, C.c_longlong, C.c_int, C.c_int, _f32p, C.c_float,
                                    C.c_int, C.c_void_p]),
    "buddy_conv3x3": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_winograd_transform_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "buddy_winograd4_transform_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "buddy_conv3x3_winograd4": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_winograd6_transform_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "buddy_conv3x3_winograd6": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_gn_conv3x3_winograd6": (C.c_int, [_f32p, _f32p, C.c_int, _f32p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_conv3x3_winograd6_gn_bwd_sums": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int,
                                                      C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_gemm_bf16x3_gn_bwd": (C.c_int, [_f32p, C.c_int, C.c_void_p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_float,
                                           _f32p, _f32p, C.c_int, C.c_int, C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_gemm_f16x2": (C.c_int, [_f32p, C.c_int, _f32p, C.c_int, C.c_int, C.c_void_p, _f32p, C.c_int, C.c_longlong, C.c_int, C.c_int, _f32p, C.c_float,
                                    C.c_int, C.c_void_p]),
    "buddy_gemm_f16x2_gn_bwd": (C.c_int, [_f32p, C.c_int, C.c_void_p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_float,
                                           _f32p, _f32p, C.c_int, C.c_int, C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_gn_upconv3x3_winograd6": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_gnbwd_upconv3x3_winograd6": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_void_p, _f32p,
                                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_gnbwd_conv3x3_winograd6": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_void_p, _f32p,
                                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_gn_conv3x3_winograd4": (C.c_int, [_f32p, _f32p, C.c_int, _f32p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_conv3x3_winograd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_groupnorm_act": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]),
    "buddy_groupnorm_act_bwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "buddy_fir_resample2": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "buddy_ncsnpp_set_fir": (C.c_int, [C.c_void_p, C.c_int]),
    "buddy_ncsnpp_set_gemm": (C.c_int, [C.c_void_p, C.c_int]),
    "buddy_ncsnpp_set_attention": (C.c_int, [C.c_void_p, C.c_int]),
    "buddy_options_check": (C.c_int, []),
    "buddy_option_validate": (C.c_int, [C.c_char_p, C.c_int]),
    "buddy_ncsnpp_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "buddy_ncsnpp_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "buddy_flash_attention_fwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "buddy_flash_attention_bwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float,
                                            C.c_int, C.c_void_p]),
    "buddy_flash_attention16_workspace": (C.c_longlong, [C.c_int, C.c_int, C.c_int]),
    "buddy_flash_attention16_fwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _f32p, C.c_void_p]),
    "buddy_flash_attention16_bwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float,
                                              C.c_int, _f32p, C.c_void_p]),
    "buddy_flash_attention_splits": (C.c_int, [C.c_int, C.c_int]),
    "buddy_flash_attention_workspace": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "buddy_flash_attention_fwd_split": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _f32p, C.c_void_p]),
    "buddy_flash_attention_bwd_split": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float,
                                                  C.c_int, _f32p, C.c_void_p]),
    "buddy_axpby_rows": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_void_p]),
    "buddy_perturb": (C.c_int, [_f32p, _f32p, C.c_float, _f32p, C.c_longlong, C.c_void_p]),
    "buddy_dps_update": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_float, _f32p, _f32p, _f32p,
                                   C.c_int, C.c_int, C.c_void_p]),
    "buddy_row_scale": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "buddy_fill_rows4": (C.c_int, [_f32p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "buddy_row_moments": (C.c_int, [_f32p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "buddy_blindop_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "buddy_blindop_destroy": (C.c_int, [C.c_void_p]),
    "buddy_blindop_set_params": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_void_p]),
    "buddy_blindop_get_params": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_update_H": (C.c_int, [C.c_void_p, _f32p, C.c_void_p]),
    "buddy_blindop_get_H": (C.c_int, [C.c_void_p, _f32p, C.c_void_p]),
    "buddy_blindop_set_y": (C.c_int, [C.c_void_p, _f32p, C.c_void_p]),
    "buddy_blindop_degrade": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_time_rir": (C.c_int, [C.c_void_p, _f32p, C.c_void_p]),
    "buddy_blindop_design_filter": (C.c_int, [C.c_void_p, _f32p, C.c_void_p]),
    "buddy_blindop_apply_stft": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_degrade_vjp": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_time_rir_vjp": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_update_H_vjp": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_stft": (C.c_int, [C.c_void_p, _f32p, C.c_int, _f32p, C.c_void_p]),
    "buddy_blindop_stft_adjoint": (C.c_int, [C.c_void_p, _f32p, C.c_int, _f32p, C.c_void_p]),
    "buddy_blindop_stft_loss": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int, C.c_float, _f32p, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_set_compression": (C.c_int, [C.c_void_p, C.c_float]),
    "buddy_blindop_set_loss_norm": (C.c_int, [C.c_void_p, C.c_int]),
    "buddy_blindop_lengths": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "buddy_blindop_minphase": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_project": (C.c_int, [C.c_void_p, C.c_void_p]),
    "buddy_blindop_get_adam": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.POINTER(C.c_int), C.c_void_p]),
    "buddy_blindop_rec_loss_grad": (C.c_int, [C.c_void_p, _f32p, C.c_float, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_fir_loss_grad": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_longlong, C.c_int, C.c_float, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_param_grads": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_float, C.c_float, _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "buddy_blindop_optimize": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_void_p]),
    "buddy_fir": (C.c_int, [_f32p, _f32p, C.c_longlong, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}
EXPORTED = sorted(_SIGS)


def load():
    """Load the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BuddyHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(buddy_amd/csrc/build.sh); there is no CPU fallback")
        import torch  # noqa: F401  -- FIRST: torch ships its own libamdhip64 (same SONAME); loading ours before it would put two
        #                       HIP runtimes in the process ("no ROCm-capable device is detected" from the second one)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
        rc = lib.buddy_options_check()     # a misspelt BUDDY_* switch fails here, loudly, not somewhere inside a launcher
        if rc != 0:
            msg = lib.buddy_last_error().decode()
            _lib = None
            raise BuddyHipError(f"libbuddy_hip: {msg}")
    return _lib


def check(rc):
    if rc != 0:
        raise BuddyHipError(f"libbuddy_hip error {rc}: {load().buddy_last_error().decode()}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise BuddyHipError("no HIP device visible: the MI355X path has no CPU fallback")
    return load()


def ptr(t):
    """device pointer of a contiguous float32 CUDA tensor (or None)."""
    if t is None:
        return None
    import torch
    assert t.is_cuda and t.is_contiguous(), "expected a contiguous device tensor"
    assert t.dtype in (torch.float32, torch.float64)
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
