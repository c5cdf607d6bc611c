"""ctypes binding of libveex.so (the C-ABI declared in include/veex.h).

The product path has no CPU fallback: if the HIP library is missing or a symbol cannot be
resolved, importing this module raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C vae_eeg_xai_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libveex.so")

c_float_p = C.c_void_p   # device pointers travel as integers
c_stream = C.c_void_p


class VeexError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    """struct VeexGemmDesc (include/veex.h)."""
    _fields_ = [
        ("inp", C.c_void_p), ("weight", C.c_void_p), ("out", C.c_void_p),
        ("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("bias", C.c_void_p),
        ("aux", C.c_void_p), ("e_scale", C.c_void_p), ("partials", C.c_void_p),
        ("N", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32), ("Cout", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_begin", C.c_int32),
        ("mode", C.c_int32), ("affine_channels", C.c_int32), ("affine_spatial", C.c_int32),
        ("epilogue", C.c_int32), ("relu", C.c_int32), ("e_channels", C.c_int32), ("nsplit", C.c_int32),
        ("in_bf16", C.c_void_p), ("out_bf16", C.c_void_p), ("out_scale", C.c_void_p), ("out_shift", C.c_void_p),
        ("out_channels", C.c_int32), ("struct_size", C.c_int32),
        ("weight_bf16", C.c_void_p), ("weight_t", C.c_void_p), ("out_bn", C.c_void_p), ("weight_planes", C.c_int32),
    ]


class WgradDesc(C.Structure):
    """struct VeexWgradDesc (include/veex.h)."""
    _fields_ = [
        ("big", C.c_void_p), ("small", C.c_void_p), ("out", C.c_void_p),
        ("big_scale", C.c_void_p), ("big_shift", C.c_void_p),
        ("small_scale", C.c_void_p), ("small_shift", C.c_void_p),
        ("big_affine_channels", C.c_int32), ("big_affine_spatial", C.c_int32),
        ("small_affine_channels", C.c_int32), ("small_affine_spatial", C.c_int32),
        ("N", C.c_int32), ("Hb", C.c_int32), ("Wb", C.c_int32), ("Cb", C.c_int32),
        ("Hs", C.c_int32), ("Ws", C.c_int32), ("Cs", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_begin", C.c_int32),
        ("nsplit", C.c_int32), ("rows_per_split", C.c_int32), ("bf16", C.c_int32),
        ("big_bf16", C.c_void_p), ("small_bf16", C.c_void_p),
    ]


_P, _I, _L, _F, _U, _D = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_ulonglong, C.c_double

# name -> (restype, argtypes); every symbol include/veex.h declares
SIGNATURES = {
    "veex_last_error": (C.c_char_p, []),
    "veex_abi_version": (_I, []),
    "veex_gather_gemm": (_I, [C.POINTER(GemmDesc), _P]),
    "veex_gather_gemm_m_tiles": (_I, [C.POINTER(GemmDesc)]),
    "veex_pack_weights_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "veex_transpose_weights_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "veex_net_set_compute": (_I, [_P, _I]),
    "veex_net_set_side_stream": (_I, [_P, _P]),
    "veex_net_set_batch_divisor": (_I, [_P, _D]),
    "veex_wgrad_gemm": (_I, [C.POINTER(WgradDesc), _P]),
    "veex_wgrad_plan": (_I, [C.POINTER(WgradDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "veex_conv1_fwd": (_I, [_P, _P, _P, _P, _I, _P]),
    "veex_conv1_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "veex_conv1_fwd_copies": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "veex_conv1_wgrad_blocks": (_I, [_I]),
    "veex_conv1_wgrad": (_I, [_P, _P, _P, _I, _P]),
    "veex_conv1_dgrad": (_I, [_P, _P, _P, _P, _I, _P]),
    "veex_deconv6_bce_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P]),
    "veex_deconv6_bce_fwd_form": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _I, _P]),
    "veex_net_set_semantics": (_I, [_P, _I, _I]),
    "veex_pack_weights_bf16x3": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "veex_net_set_sample_counter": (_I, [_P, _P]),
    "veex_net_set_bucket_event": (_I, [_P, _P]),
    "veex_net_grad_bucket": (_I, [_P, C.POINTER(_L), C.POINTER(_L)]),
    "veex_reparam_kl_fwd_ctr": (_I, [_P, _P, _P, _P, _I, _I, _I, _U, _U, _P, _P]),
    "veex_counter_add": (_I, [_P, _L, _P]),
    "veex_deconv6_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "veex_deconv6_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "veex_reparam_kl_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _U, _U, _P]),
    "veex_reparam_kl_bwd": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "veex_bn_prepare": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _F, _P]),
    "veex_bnrelu_bwd_rows_chunks": (_I, [_I]),
    "veex_bnrelu_bwd_rows": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "veex_colsum_finalize": (_I, [_P, _I, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "veex_slab_sum": (_I, [_P, _I, _L, _L, _P, _P, _I, _I, _P]),
    "veex_adam": (_I, [_P, _P, _P, _P, _L, _P, _P, _D, _D, _D, _D, _D, _P]),
    "veex_minmax_partial_blocks": (_I, [_L]),
    "veex_minmax": (_I, [_P, _L, _P, _P, _P]),
    "veex_minmax_static": (_I, [_P, _L, _P, _P]),
    "veex_normalize": (_I, [_P, _P, _P, _L, _P]),
    "veex_denormalize": (_I, [_P, _P, _P, _L, _P]),
    "veex_electrode_readout": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "veex_column_quantiles_scratch_bytes": (_L, [_I, _I]),
    "veex_column_quantiles": (_I, [_P, _I, _I, C.POINTER(_D), _I, _P, _P, _P]),
    "veex_column_medians": (_I, [_P, _I, _I, _P, _P, _P]),
    "veex_band_outside_counts": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "veex_net_create": (_P, [_I, _I]),
    "veex_net_destroy": (None, [_P]),
    "veex_net_param_floats": (_L, [_P]),
    "veex_net_state_floats": (_L, [_P]),
    "veex_net_num_tensors": (_I, [_P]),
    "veex_net_tensor_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_I), C.POINTER(_L),
                                  C.POINTER(_I), C.POINTER(_I)]),
    "veex_net_workspace_floats": (_L, [_P, _I]),
    "veex_net_ws_offset": (_L, [_P, _I, C.c_char_p, C.POINTER(_L)]),
    "veex_net_encode": (_I, [_P, _P, _P, _P, _P, _I, _I, _U, _U, _P, _L, _P, _P, _P, _P]),
    "veex_net_decode": (_I, [_P, _P, _P, _P, _I, _P, _L, _P, _P, _P]),
    "veex_net_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _U, _U, _P, _L, _P, _P, _P, _P]),
    "veex_net_train_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _U, _U, _P, _L, _P, _P, _P, _P]),
    "veex_net_attribute": (_I, [_P, _P, _P, _P, _I, C.POINTER(_I), _I, _I, _P, _L, _P, _P]),
}

ABI_VERSION = 3               # VEEX_ABI_VERSION of include/veex.h
F_PHILOX = 1
F_ZMEAN_DECODE = 2
VARIANT_VAE, VARIANT_AE = 0, 1
COMPUTE_F32, COMPUTE_BF16, COMPUTE_F32X3 = 0, 1, 2


def _load():
    if not os.path.exists(LIB_PATH):
        raise VeexError(
            f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.veex_abi_version() != ABI_VERSION:   # descriptor structs below mirror ONE version of include/veex.h
        raise VeexError(f"{LIB_PATH} has ABI version {lib.veex_abi_version()}, this binding is for {ABI_VERSION}: rebuild it")
    return lib


lib = _load()


def setenv(name: str, value) -> None:
    """Flip a tuning switch inside a running process: libveex.so freezes every switch at its
    first use, so the A/B tools set the variable AND make the library forget its cached values.
    ``value=None`` unsets it (back to the default)."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    lib.veexi_debug_reload_env()


def hg_plan_code(G: int = 0, bands: int = 0, ch: int = 0, nt: int = 0, occ: int = 0) -> int:
    """Value of VEEX_HG_PLAN that forces the halo-tile kernel's plan (csrc/halo_gemm_bf16.hip hg_forced): samples per
    workgroup, bands of output rows, channel slice (multiple of 32), 64-column sub-tiles, waves per SIMD; 0 = search."""
    assert 0 <= bands < 10 and ch % 32 == 0 and 0 <= ch // 32 < 100 and 0 <= nt < 10 and 0 <= occ < 10
    return G * 100000 + bands * 10000 + (ch // 32) * 100 + nt * 10 + occ


def check(rc: int, what: str = "veex call"):
    if rc != 0:
        raise VeexError(f"{what} failed (rc={rc}): {lib.veex_last_error().decode()}")


def ptr(t):
    """Device (or host) pointer of a torch tensor, or None."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
