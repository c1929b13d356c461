"""ctypes binding of libparrot_hip.so (see include/parrot_hip.h).

The HIP library is the product path.  There is deliberately no CPU fallback here: if the shared
object is missing or a call fails, we raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libparrot_hip.so")

MAX_LAYERS = 3

c_float_p = C.c_void_p  # device pointers are passed as integers


class HipLibraryMissing(RuntimeError):
    pass


class HipCallError(RuntimeError):
    pass


class GruSeqDesc(C.Structure):
    _fields_ = [
        ("T", C.c_int), ("B", C.c_int), ("H", C.c_int), ("nchain", C.c_int),
        ("use_graph", C.c_int), ("reserved", C.c_int),
        ("reverse", C.c_int * 4),
        ("Wg", C.c_void_p * 4), ("Wc", C.c_void_p * 4),
        ("inputs", C.c_void_p * 4), ("gate_inputs", C.c_void_p * 4),
        ("mask", C.c_void_p),
        ("h", C.c_void_p * 4),
        ("z", C.c_void_p * 4), ("r", C.c_void_p * 4), ("rh", C.c_void_p * 4), ("c", C.c_void_p * 4),
        ("dh", C.c_void_p * 4), ("dG", C.c_void_p * 4), ("dC", C.c_void_p * 4),
    ]


class LstmSeqDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("T", "B", "H", "use_graph")] + \
               [(n, C.c_void_p) for n in ("W", "pre_in", "s", "c", "gates", "dS", "dc", "dP")]


class DecoderDesc(C.Structure):
    _fields_ = [
        ("T", C.c_int), ("B", C.c_int), ("H", C.c_int), ("E", C.c_int), ("A", C.c_int),
        ("U", C.c_int), ("L", C.c_int), ("att_type", C.c_int), ("use_graph", C.c_int),
        ("reserved", C.c_int),
        ("eps", C.c_float), ("alignment", C.c_float), ("sharpening", C.c_float), ("timing", C.c_float),
        ("Wg", C.c_void_p * MAX_LAYERS), ("Wc", C.c_void_p * MAX_LAYERS),
        ("bg", C.c_void_p * MAX_LAYERS), ("bc", C.c_void_p * MAX_LAYERS),
        ("WattT", C.c_void_p), ("batt", C.c_void_p), ("ctx", C.c_void_p),
        ("seq_c", C.c_void_p * MAX_LAYERS), ("seq_g", C.c_void_p * MAX_LAYERS),
        ("h", C.c_void_p * MAX_LAYERS), ("w", C.c_void_p), ("kappa", C.c_void_p),
        ("z", C.c_void_p * MAX_LAYERS), ("r", C.c_void_p * MAX_LAYERS),
        ("rh", C.c_void_p * MAX_LAYERS), ("c", C.c_void_p * MAX_LAYERS),
        ("a", C.c_void_p), ("b", C.c_void_p), ("phi", C.c_void_p),
        ("dh", C.c_void_p * MAX_LAYERS), ("dw", C.c_void_p), ("dw0", C.c_void_p), ("dhup", C.c_void_p * MAX_LAYERS),
        ("dkappa", C.c_void_p),
        ("dG", C.c_void_p * MAX_LAYERS), ("dC", C.c_void_p * MAX_LAYERS), ("dp", C.c_void_p),
        ("cell", C.c_int), ("seq_init", C.c_int),
        ("cst", C.c_void_p * MAX_LAYERS), ("gate4", C.c_void_p * MAX_LAYERS), ("dcell", C.c_void_p * MAX_LAYERS),
        ("layer_norm", C.c_int), ("bf16", C.c_int),
        ("ln_bg", C.c_void_p * (MAX_LAYERS * MAX_LAYERS)), ("ln_bc", C.c_void_p * (MAX_LAYERS * MAX_LAYERS)),
        ("ln_yg", C.c_void_p * (MAX_LAYERS * MAX_LAYERS)), ("ln_yc", C.c_void_p * (MAX_LAYERS * MAX_LAYERS)),
        ("ln_sg", C.c_void_p * (MAX_LAYERS * MAX_LAYERS)), ("ln_sc", C.c_void_p * (MAX_LAYERS * MAX_LAYERS)),
        ("Wg_f", C.c_void_p * MAX_LAYERS), ("Wc_f", C.c_void_p * MAX_LAYERS),
        ("Wg_r", C.c_void_p * MAX_LAYERS), ("Wc_r", C.c_void_p * MAX_LAYERS),
        ("att_sup", C.c_void_p),
        ("persist_ws", C.c_void_p), ("persist_ws_floats", C.c_longlong),
        ("dh_b", C.c_void_p * MAX_LAYERS), ("dhup_b", C.c_void_p * MAX_LAYERS), ("dw_b", C.c_void_p), ("dw0_b", C.c_void_p),
        ("dhup_c", C.c_void_p * MAX_LAYERS), ("dw_c", C.c_void_p), ("dw0_c", C.c_void_p),
        ("dG16", C.c_void_p * MAX_LAYERS),
    ]


class SampleDesc(C.Structure):
    _fields_ = [
        ("S", C.c_int), ("B", C.c_int), ("H", C.c_int), ("E", C.c_int), ("A", C.c_int),
        ("U", C.c_int), ("L", C.c_int), ("O", C.c_int), ("R", C.c_int), ("ldx", C.c_int),
        ("att_type", C.c_int), ("use_graph", C.c_int),
        ("eps", C.c_float), ("alignment", C.c_float), ("sharpening", C.c_float), ("timing", C.c_float),
        ("Wg", C.c_void_p * MAX_LAYERS), ("Wc", C.c_void_p * MAX_LAYERS),
        ("bg", C.c_void_p * MAX_LAYERS), ("bc", C.c_void_p * MAX_LAYERS),
        ("Wfg", C.c_void_p * MAX_LAYERS), ("Wfc", C.c_void_p * MAX_LAYERS),
        ("seq_c", C.c_void_p * MAX_LAYERS), ("seq_g", C.c_void_p * MAX_LAYERS),
        ("WattT", C.c_void_p), ("batt", C.c_void_p),
        ("Wr", C.c_void_p), ("br", C.c_void_p), ("radd", C.c_void_p),
        ("Wo", C.c_void_p), ("bo", C.c_void_p), ("oadd", C.c_void_p),
        ("ctx", C.c_void_p),
        ("x", C.c_void_p), ("h", C.c_void_p * MAX_LAYERS), ("w", C.c_void_p), ("kappa", C.c_void_p),
        ("a", C.c_void_p), ("bwork", C.c_void_p), ("phi", C.c_void_p),
        ("zwork", C.c_void_p), ("rwork", C.c_void_p), ("rhwork", C.c_void_p), ("readout", C.c_void_p),
        ("gmm_K", C.c_int), ("reserved2", C.c_int), ("sampling_bias", C.c_float), ("reserved3", C.c_float),
        ("Wmu", C.c_void_p), ("bmu", C.c_void_p), ("Wsig", C.c_void_p), ("bsig", C.c_void_p),
        ("Wco", C.c_void_p), ("bco", C.c_void_p),
        ("add_mu", C.c_void_p), ("add_sig", C.c_void_p), ("add_co", C.c_void_p),
        ("unif", C.c_void_p), ("noise", C.c_void_p),
        ("gmm_mu", C.c_void_p), ("gmm_sig", C.c_void_p), ("gmm_co", C.c_void_p), ("pi_out", C.c_void_p),
        ("cell", C.c_int), ("reserved5", C.c_int),
        ("cwork", C.c_void_p * MAX_LAYERS), ("gwork", C.c_void_p),
        ("layer_norm", C.c_int), ("reserved7", C.c_int),
        ("bfg", C.c_void_p * MAX_LAYERS), ("bfc", C.c_void_p * MAX_LAYERS),
        ("ln_bg", C.c_void_p * (MAX_LAYERS * MAX_LAYERS)), ("ln_bc", C.c_void_p * (MAX_LAYERS * MAX_LAYERS)),
        ("br_l", C.c_void_p * MAX_LAYERS),
        ("ln_scratch", C.c_void_p), ("ln_scratch_floats", C.c_longlong),
        ("Wg_t", C.c_void_p * MAX_LAYERS), ("Wc_t", C.c_void_p * MAX_LAYERS),
        ("Wr_t", C.c_void_p), ("Wo_t", C.c_void_p), ("bo_pad", C.c_void_p), ("oadd_pad", C.c_void_p),
        ("persist_ws", C.c_void_p), ("persist_ws_floats", C.c_longlong),
        ("Wro_t", C.c_void_p), ("ro_const", C.c_void_p),
        ("Wgx_t", C.c_void_p * MAX_LAYERS), ("Wcx_t", C.c_void_p * MAX_LAYERS),
        ("Watt_t", C.c_void_p),
    ]


class SampleRnnGenDesc(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("B", "D", "T", "Q", "FS", "BFS", "feat_dim", "use_graph")] +
                [("temperature", C.c_float), ("reserved", C.c_int), ("seed", C.c_ulonglong)] +
                [(n, C.c_void_p) for n in (
                    "big_Win_frames", "big_Win_feats", "big_bin", "big_U", "big_bU", "big_Wg", "big_Wc",
                    "big_Wout", "big_bout", "frm_Win", "frm_bin", "frm_U", "frm_bU", "frm_Wg", "frm_Wc",
                    "frm_Wout", "frm_bout", "emb_tbl", "W2", "b2", "W3", "b3", "W4", "b4", "features",
                    "samples", "big_h", "frm_h", "xf_big", "xf_frm", "feat_cur", "gru_in", "P", "z", "r", "rh",
                    "big_out", "frame_out", "o1", "o2", "o3", "logits", "tbase")] +
                [("n_rnn", C.c_int), ("lstm", C.c_int),
                 ("big_L", (C.c_void_p * 4) * 5), ("frm_L", (C.c_void_p * 4) * 5),
                 ("big_hs", C.c_void_p * 5), ("big_cs", C.c_void_p * 5),
                 ("frm_hs", C.c_void_p * 5), ("frm_cs", C.c_void_p * 5),
                 ("gate_ws", C.c_void_p), ("layer_tmp", C.c_void_p),
                 ("persist_ws", C.c_void_p), ("persist_ws_floats", C.c_longlong)])


# name -> (restype, argtypes); every symbol include/parrot_hip.h declares must be listed here
# (tests/test_capi_cpu.py cross-checks this table against the header).
_vp, _i, _f, _ll, _sz = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t
SIGNATURES = {
    "parrot_hip_version": (C.c_char_p, []),
    "parrot_profile_begin": (_i, []),
    "parrot_profile_end": (C.c_longlong, [C.POINTER(C.c_double)] * 3),
    "parrot_profile_end2": (C.c_longlong, [C.POINTER(C.c_double)] * 4),
    "parrot_gemm": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _f, _i, _i, _i, _ll, _ll, _ll, _i, _vp]),
    "parrot_gemm_gated": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "parrot_gather_sum_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _ll, _i, _i, _i, _vp]),
    "parrot_gather_sum_bwd_ws_floats": (C.c_longlong, [_ll, _i, _i, _i]),
    "parrot_gather_sum_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _ll, _ll, _i, _i, _i, _i, _vp]),
    "parrot_softmax_ce_fwd": (_i, [_vp, _i, _vp, _ll, _i, _vp, _vp, _vp]),
    "parrot_softmax_ce_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _ll, _i, _vp, _i, _vp]),
    "parrot_relu_gate": (_i, [_vp, _vp, _vp, _ll, _vp]),
    "samplernn_weightnorm_ws_floats": (_ll, [_i]),
    "samplernn_weightnorm_fold": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "samplernn_weightnorm_fold_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "parrot_colsum": (_i, [_vp, _ll, _i, _i, _vp, _i, _vp]),
    "parrot_gru_step_fwd": (_i, [_vp] * 11 + [_i, _i, _vp]),
    "parrot_gru_step_bwd": (_i, [_vp] * 11 + [_i, _i, _vp]),
    "parrot_gru_seq_create": (_i, [C.POINTER(GruSeqDesc), C.POINTER(C.c_void_p)]),
    "parrot_gru_seq_fwd": (_i, [_vp, _vp]),
    "parrot_gru_seq_bwd": (_i, [_vp, _vp]),
    "parrot_gru_seq_destroy": (_i, [_vp]),
    "parrot_lstm_seq_create": (_i, [C.POINTER(LstmSeqDesc), C.POINTER(C.c_void_p)]),
    "parrot_lstm_seq_fwd": (_i, [_vp, _vp]),
    "parrot_lstm_seq_bwd": (_i, [_vp, _vp]),
    "parrot_lstm_seq_destroy": (_i, [_vp]),
    "parrot_gmm_attention_fwd": (_i, [_vp] * 10 + [_i] * 6 + [_f] * 4 + [_vp]),
    "parrot_gmm_attention_bwd": (_i, [_vp] * 10 + [_i] * 6 + [_f, _vp]),
    "parrot_decoder_create": (_i, [C.POINTER(DecoderDesc), C.POINTER(C.c_void_p)]),
    "parrot_decoder_persist_floats": (C.c_longlong, [C.POINTER(DecoderDesc)]),
    "parrot_decoder_is_persistent": (_i, [_vp]),
    "parrot_decoder_schedule": (_i, [_vp]),
    "parrot_decoder_backward_tick": (_i, [_vp]),
    "parrot_decoder_writes_bf16_grads": (_i, [_vp]),
    "parrot_decoder_trace": (C.c_longlong, [_vp, _i, C.POINTER(C.c_longlong), C.c_longlong]),
    "parrot_decoder_trace_jobs": (C.c_longlong, [_vp, _i, C.POINTER(C.c_longlong), C.c_longlong]),
    "parrot_decoder_seq_fwd": (_i, [_vp, _vp]),
    "parrot_decoder_seq_bwd": (_i, [_vp, _vp]),
    "parrot_decoder_destroy": (_i, [_vp]),
    "parrot_sample_create": (_i, [C.POINTER(SampleDesc), C.POINTER(C.c_void_p)]),
    "parrot_sample_persist_floats": (C.c_longlong, [C.POINTER(SampleDesc)]),
    "parrot_sample_is_persistent": (_i, [_vp]),
    "parrot_sample_plan_pieces_dry": (_i, [C.POINTER(SampleDesc), _i, C.POINTER(C.c_int)]),
    "parrot_sample_status": (_i, [_vp]),
    "parrot_decoder_status": (_i, [_vp]),
    "parrot_sample_run": (_i, [_vp, _vp]),
    "parrot_sample_destroy": (_i, [_vp]),
    "parrot_plan_last_error": (_i, [_vp]),
    "parrot_sumsq": (_i, [_vp, _sz, _vp, _vp]),
    "parrot_tile_weights": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "parrot_tile_weights_bf16": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "parrot_set_gemm_precision": (_i, [_i]),
    "parrot_get_gemm_precision": (_i, []),
    "parrot_to_bf16": (_i, [_vp, _vp, _ll, _vp]),
    "parrot_gemm_bf16in": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "parrot_gemm_bf16in_ex": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "parrot_simple_norm_fwd": (_i, [_vp, _i, _vp, _i, _vp, _ll, _i, _f, _vp, _i, _vp]),
    "parrot_simple_norm_bwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _ll, _i, _f, _i, _vp]),
    "parrot_adam_clip_step": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _f, _f, _f, _f, _f, _f, _i, _vp]),
    "parrot_batch_quantize": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    "parrot_mu2linear": (_i, [_vp, _sz, _vp, _vp]),
    "samplernn_generate_create": (_i, [C.POINTER(SampleRnnGenDesc), C.POINTER(C.c_void_p)]),
    "samplernn_generate_run": (_i, [_vp, _vp]),
    "samplernn_persist_floats": (C.c_longlong, [C.POINTER(SampleRnnGenDesc)]),
    "samplernn_generate_is_persistent": (_i, [_vp]),
    "samplernn_generate_status": (_i, [_vp]),
    "samplernn_generate_destroy": (_i, [_vp]),
}

_lib = None


def load():
    """Returns the loaded library; raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64; it has to be in the process BEFORE this library is dlopen'ed, so that
    # the loader binds our DT_NEEDED libamdhip64.so to that one.  Loaded the other way round the process ends up
    # with two HIP runtimes and our launches on torch's pointers fail with hipErrorNoDevice (100).
    import torch  # noqa: F401
    path = os.environ.get('PARROT_HIP_LIB', LIB_PATH)  # development knob: alternative builds of the same ABI
    if not os.path.exists(path):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m parrot_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str, plan=None):
    if rc != 0:
        raise HipCallError(f"{what} failed with code {rc}")


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    check(rc, name)
    return rc
