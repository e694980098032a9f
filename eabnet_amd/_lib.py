"""ctypes binding of libeabnet_hip.so (include/eabnet_hip.h).

There is deliberately NO fallback: if the shared library is missing or does not
export the expected ABI, importing the product path raises.  A CPU/PyTorch
fallback would silently void every parity and performance claim.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libeabnet_hip.so")
ABI_VERSION = 8
MAX_TAPS = 16
_fp = C.POINTER(C.c_float)


class TimeWindow(C.Structure):
    """mirror of eab_time_window: device pointer to the current frame position + frames per chunk"""
    _fields_ = [("pos", C.c_void_p), ("count", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("src0", C.c_void_p), ("src1", C.c_void_p), ("xf0", C.c_void_p), ("xf1", C.c_void_p),
        ("slope0", C.c_void_p), ("slope1", C.c_void_p),
        ("C0", C.c_int32), ("C1", C.c_int32), ("xf_mode", C.c_int32),
        ("w", C.c_void_p), ("bias", C.c_void_p),
        ("N", C.c_int32), ("Kpad", C.c_int32),
        ("B", C.c_int32), ("T", C.c_int32), ("Fin", C.c_int32), ("Fout", C.c_int32),
        ("No", C.c_int32), ("ostride", C.c_int32), ("ophase", C.c_int32), ("istride", C.c_int32),
        ("ntaps", C.c_int32), ("dt", C.c_int32 * MAX_TAPS), ("ioff", C.c_int32 * MAX_TAPS),
        ("epi", C.c_int32),
        ("aux", C.c_void_p), ("dst", C.c_void_p), ("dst_acc", C.c_void_p),
        ("Cout", C.c_int32),
        ("stats", C.c_void_p), ("nsets", C.c_int32),
        ("stat_slope0", C.c_void_p), ("stat_slope1", C.c_void_p),
        ("stat_tiles", C.c_int32), ("stat_tile0", C.c_int32), ("bm", C.c_int32),
        ("fin_stats", C.c_void_p), ("fin_gamma0", C.c_void_p), ("fin_beta0", C.c_void_p),
        ("fin_gamma1", C.c_void_p), ("fin_beta1", C.c_void_p),
        ("fin_tiles", C.c_int32), ("fin_nsets", C.c_int32), ("fin_count", C.c_int32), ("fin_eps", C.c_float),
        ("precision", C.c_int32), ("korder", C.c_int32),
        ("win", TimeWindow),
        ("fz_counter", C.c_void_p), ("fz_gamma0", C.c_void_p), ("fz_beta0", C.c_void_p), ("fz_xf0", C.c_void_p),
        ("fz_gamma1", C.c_void_p), ("fz_beta1", C.c_void_p), ("fz_xf1", C.c_void_p), ("fz_eps", C.c_float),
        ("glu_dump", C.c_void_p),
        ("ph1_w", C.c_void_p), ("ph1_No", C.c_int32), ("ph1_ophase", C.c_int32), ("ph1_ntaps", C.c_int32), ("ph1_Kpad", C.c_int32),
        ("ph1_dt", C.c_int32 * MAX_TAPS), ("ph1_ioff", C.c_int32 * MAX_TAPS),
        ("f2_w", C.c_void_p), ("f2_dst", C.c_void_p), ("f2_stats", C.c_void_p), ("f2_stat_slope0", C.c_void_p),
        ("f2_stat_slope1", C.c_void_p), ("f2_N", C.c_int32), ("f2_nsets", C.c_int32), ("f2_stat_tiles", C.c_int32),
        ("p2_mask1", C.c_int32), ("src_bf16", C.c_int32),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("dz", C.c_void_p), ("src0", C.c_void_p), ("src1", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p),
        ("N", C.c_int32), ("C0", C.c_int32), ("C1", C.c_int32), ("Kpad", C.c_int32),
        ("B", C.c_int32), ("T", C.c_int32), ("Fin", C.c_int32), ("Fz", C.c_int32), ("No", C.c_int32),
        ("ostride", C.c_int32), ("ophase", C.c_int32), ("istride", C.c_int32),
        ("ntaps", C.c_int32), ("dt", C.c_int32 * MAX_TAPS), ("ioff", C.c_int32 * MAX_TAPS),
        ("rows_per_wg", C.c_int32), ("precision", C.c_int32), ("bf16_mask", C.c_int32),
    ]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("i", C.c_int32 * 8), ("f", C.c_float * 2), ("p", C.c_void_p * 12),
                ("win", TimeWindow), ("conv", ConvDesc), ("wgrad", WgradDesc)]


class EabError(RuntimeError):
    pass


_SIGS = {
    "eab_abi_version": (C.c_int, []),
    "eab_error_string": (C.c_char_p, [C.c_int]),
    "eab_stft_compress_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]),
    "eab_stft_frames_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 5 + [C.c_void_p]),
    "eab_filter_sum_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "eab_istft_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "eab_norm_act_win_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [TimeWindow, C.c_void_p]),
    "eab_lstm64_stream_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 4 + [C.c_int] * 4
                              + [TimeWindow, C.c_void_p]),
    "eab_bfw_filter_sum_win_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [TimeWindow, C.c_void_p]),
    "eab_mlp_bfw_filter_sum_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 4 + [TimeWindow, C.c_void_p]),
    "eab_zero_rows_f32": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [TimeWindow, C.c_void_p]),
    "eab_com_mag_mse_loss_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "eab_gag_pack_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [TimeWindow, C.c_void_p]),
    "eab_gag_crm_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 6 + [TimeWindow, C.c_void_p]),
    "eab_gag_crm_bwd_f32": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 6 + [C.c_void_p]),
    "eab_conv_tiles": (C.c_int, [C.c_int] * 3),
    "eab_conv_f32": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "eab_train_cln_bwd_f32": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 5 + [C.c_void_p]),
    "eab_conv_st_chain_plan": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "eab_conv_st_chain_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "eab_conv_bf16": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "eab_lstm64_bf16": (C.c_int, [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "eab_in_finalize_f32": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_float] + [C.c_void_p] * 6 + [C.c_void_p]),
    "eab_norm_act_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 3 + [C.c_void_p]),
    "eab_lstm64_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "eab_lstm64_prec_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "eab_bfw_filter_sum_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]),
    "eab_run_program": (C.c_int, [C.POINTER(Op), C.c_int, C.c_void_p]),
    "eab_sizeof_conv_desc": (C.c_int, []),
    "eab_sizeof_op": (C.c_int, []),
    "eab_sizeof_wgrad_desc": (C.c_int, []),
    # training (include/eabnet_hip.h, "Training")
    "eab_gather_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong, C.c_void_p]),
    "eab_train_in_stats_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_float] + [C.c_void_p] * 4 + [C.c_void_p]),
    "eab_train_in1d_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_float] + [C.c_void_p] * 5 + [C.c_void_p]),
    "eab_train_in1d_multi_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_float] + [C.c_void_p] * 5 + [C.c_void_p]),
    "eab_train_norm_bwd_multi_f32": (C.c_int, [C.c_void_p] * 12 + [C.c_int] * 3 + [C.c_void_p]),
    "eab_in_finalize_mr_f32": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_float] + [C.c_void_p] * 8 + [C.c_void_p]),
    "eab_train_norm_act_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "eab_train_norm_bwd_f32": (C.c_int, [C.c_void_p] * 12 + [C.c_int] * 4 + [C.c_void_p]),
    "eab_glu_bwd_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_longlong, C.c_int, C.c_void_p]),
    "eab_glu_bwd_ex_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    "eab_gate_fwd_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_longlong, C.c_void_p]),
    "eab_gate_bwd_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong, C.c_void_p]),
    "eab_add_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_longlong, C.c_void_p]),
    "eab_copy_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_longlong, C.c_void_p]),
    "eab_relu_bwd_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_longlong, C.c_void_p]),
    "eab_colsum_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_longlong, C.c_int, C.c_void_p]),
    "eab_filter_sum_bwd_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]),
    "eab_filter_sum_ld_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]),
    "eab_layernorm64_fwd_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 2 + [C.c_longlong, C.c_void_p]),
    "eab_layernorm64_bwd_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_longlong, C.c_void_p]),
    "eab_lstm64_train_fwd_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]),
    "eab_lstm64_bwd_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "eab_lstm64_train_fwd_prec_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "eab_lstm64_bwd_prec_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "eab_wgrad_f32": (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    "eab_wgrad_batchable": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "eab_wgrad_batch_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "eab_cln_stats_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_float] + [C.c_void_p] * 3 + [TimeWindow, C.c_void_p]),
    "eab_cln_apply_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 5 + [TimeWindow, C.c_void_p]),
    "eab_cln_step_f32": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 5 + [C.c_float, TimeWindow, C.c_void_p]),
    "eab_gate_rows_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [TimeWindow, C.c_void_p]),
}
EXPORTS = tuple(_SIGS)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load and validate the shared library (idempotent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EabError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C eabnet_amd/csrc`). eabnet_amd has no CPU fallback by design.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EabError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.eab_abi_version() != ABI_VERSION:
        raise EabError(f"ABI mismatch: library {lib.eab_abi_version()} != binding {ABI_VERSION}")
    if lib.eab_sizeof_conv_desc() != C.sizeof(ConvDesc) or lib.eab_sizeof_op() != C.sizeof(Op) \
            or lib.eab_sizeof_wgrad_desc() != C.sizeof(WgradDesc):
        raise EabError("struct layout mismatch between include/eabnet_hip.h and eabnet_amd/_lib.py")
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().eab_error_string(code).decode()
        raise EabError(f"{what or 'libeabnet_hip'} failed with code {code}: {msg}")
