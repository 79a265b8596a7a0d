"""ctypes binding of libneuspeech_hip.so (the C ABI in include/neuspeech_hip.h).

There is deliberately NO fallback: if the shared object is missing or a symbol
is absent, importing/using the product path raises.  The oracle under oracle/
is test infrastructure and is never reachable from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NS_LIB_PATH") or os.path.join(_HERE, "libneuspeech_hip.so")   # NS_LIB_PATH: same-box A/B of two builds (probes only)


class NeuSpeechHipError(RuntimeError):
    pass


# Held while a hipGraph capture is open (engine.train_step, generate's decode loop) and by every other host thread of this
# package around its GPU API calls (the data feed's loader thread: device synchronize, pinned / device allocations, copies).
# HIP invalidates an open capture when another thread synchronizes the device or allocates meanwhile, thread-local capture
# mode notwithstanding (seen on ROCm 7.0/7.2: hipErrorStreamCaptureInvalidated at the first captured launch).
import threading  # noqa: E402
GPU_CAPTURE_LOCK = threading.RLock()

ABI_VERSION = 3      # ns_version() of the library this binding was written against (3: ns_zero_spans / ns_add_i32)


class RowMap(C.Structure):
    _fields_ = [("seg_stride", C.c_int64), ("seg_rows", C.c_int32), ("ld", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("am", RowMap), ("K", C.c_int32),
        ("B", C.c_void_p), ("bm", RowMap),
        ("A2", C.c_void_p), ("am2", RowMap), ("K2", C.c_int32),
        ("B2", C.c_void_p), ("ldb2", C.c_int32),
        ("a2_ngroup", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32),
        ("bias", C.c_void_p),
        ("C16", C.c_void_p), ("c16m", RowMap),
        ("G16", C.c_void_p), ("g16m", RowMap),
        ("P16", C.c_void_p), ("p16m", RowMap),
        ("R32", C.c_void_p), ("H32", C.c_void_p), ("h32m", RowMap),
        ("pos", C.c_void_p), ("pos_rows", C.c_int32),
        ("C32", C.c_void_p), ("ldc32", C.c_int32),
        ("flags", C.c_int32),
        ("splits", C.c_int32),
        ("drop_p", C.c_float), ("drop_seed", C.c_uint32),
        ("alpha", C.c_float),
        ("side_B", C.c_void_p), ("side_ldb", C.c_int32), ("side_n", C.c_int32),
        ("side_out", C.c_void_p),
        ("side_drop_p", C.c_float), ("side_drop_seed", C.c_uint32),
        ("seed_dev", C.c_void_p),
    ]


class GemmLnDesc(C.Structure):
    _fields_ = [("g", GemmDesc), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("ldx", C.c_int32),
                ("x16", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p)]


class Span(C.Structure):
    _fields_ = [("p", C.c_void_p), ("bytes", C.c_size_t)]


class CastJob(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p),
        ("rows", C.c_int32), ("cols", C.c_int32), ("ld_src", C.c_int32), ("ld_dst", C.c_int32),
        ("scale", C.c_float), ("transpose", C.c_int32),
        ("colscale", C.c_void_p),
    ]


class OrthJob(C.Structure):
    _fields_ = [("P", C.c_void_p), ("G", C.c_void_p), ("r", C.c_int32), ("len", C.c_int32), ("ld", C.c_int32),
                ("is_b", C.c_int32)]


class AdaloraFoldJob(C.Structure):
    _fields_ = [("dBf", C.c_void_p), ("B", C.c_void_p), ("E", C.c_void_p), ("dB", C.c_void_p), ("dE", C.c_void_p),
                ("N", C.c_int32), ("r", C.c_int32), ("s", C.c_float)]


class AttnDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
        ("dO", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
        ("LSE", C.c_void_p), ("Delta", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32), ("head_dim", C.c_int32),
        ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
        ("lddo", C.c_int32), ("lddq", C.c_int32), ("lddk", C.c_int32), ("lddv", C.c_int32),
        ("causal", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class FeedItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ld", C.c_int64), ("rows", C.c_int32), ("n", C.c_int32), ("dtype", C.c_int32),
                ("reserved", C.c_int32)]


NS_FEED_F64, NS_FEED_F32, NS_FEED_F16 = 0, 1, 2


class AdamWCfg(C.Structure):
    _fields_ = [
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("weight_decay", C.c_float), ("max_grad_norm", C.c_float),
        ("warmup_steps", C.c_int32), ("total_steps", C.c_int32),
        ("scale_growth", C.c_float), ("scale_backoff", C.c_float), ("scale_interval", C.c_int32),
    ]


class AttnDecodeDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
        ("anc", C.c_void_p), ("kv_len_dev", C.c_void_p),
        ("groups", C.c_int32), ("nq", C.c_int32), ("H", C.c_int32), ("Lk", C.c_int32), ("Lk_max", C.c_int32),
        ("head_dim", C.c_int32),
        ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32), ("anc_ld", C.c_int32),
        ("kv_group_stride", C.c_int64), ("kv_pos_stride", C.c_int64),
        ("Knew", C.c_void_p), ("Vnew", C.c_void_p), ("ldnew", C.c_int32), ("slot0", C.c_int32),
    ]


class AttnFewqDesc(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("Vt", C.c_void_p), ("O", C.c_void_p),
                ("groups", C.c_int32), ("nq", C.c_int32), ("H", C.c_int32), ("Lk", C.c_int32),
                ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldvt", C.c_int32), ("ldo", C.c_int32)]


class LogitsProcDesc(C.Structure):
    _fields_ = [
        ("logits16", C.c_void_p), ("scores32", C.c_void_p), ("ids", C.c_void_p), ("beam_scores", C.c_void_p),
        ("suppress", C.c_void_p), ("begin_suppress", C.c_void_p), ("cur_len_dev", C.c_void_p),
        ("rows", C.c_int32), ("V", C.c_int32), ("ldv", C.c_int32), ("ids_ld", C.c_int32), ("cur_len", C.c_int32),
        ("begin_index", C.c_int32),
        ("n_suppress", C.c_int32), ("n_begin_suppress", C.c_int32), ("no_repeat_ngram", C.c_int32),
        ("log_softmax", C.c_int32),
        ("repetition_penalty", C.c_float),
        ("bias1", C.c_void_p), ("seq_tok", C.c_void_p), ("seq_off", C.c_void_p), ("seq_bias", C.c_void_p),
        ("n_seq", C.c_int32), ("n_forced", C.c_int32), ("forced", C.c_void_p),
    ]


class LoraBwdDesc(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("u", C.c_void_p), ("du", C.c_void_p),
        ("sBT", C.c_void_p * 3), ("dB", C.c_void_p * 3),
        ("M", C.c_int32), ("N", C.c_int32), ("r", C.c_int32), ("G", C.c_int32),
        ("ldy", C.c_int32), ("ldu", C.c_int32), ("lddu", C.c_int32), ("lddb", C.c_int32),
        ("alpha_du", C.c_float), ("alpha_db", C.c_float * 3),
        ("splits", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class BeamDesc(C.Structure):
    _fields_ = [
        ("top_vals", C.c_void_p), ("top_idx", C.c_void_p),
        ("run_seqs_in", C.c_void_p), ("run_seqs_out", C.c_void_p), ("run_scores_out", C.c_void_p),
        ("fin_seqs_in", C.c_void_p), ("fin_seqs_out", C.c_void_p),
        ("fin_scores_in", C.c_void_p), ("fin_scores_out", C.c_void_p),
        ("fin_done_in", C.c_void_p), ("fin_done_out", C.c_void_p),
        ("open", C.c_void_p), ("parent_out", C.c_void_p), ("next_tok_out", C.c_void_p),
        ("any_open", C.c_void_p), ("any_continuation", C.c_void_p), ("cur_len_dev", C.c_void_p),
        ("batch", C.c_int32), ("num_beams", C.c_int32), ("V", C.c_int32), ("max_len", C.c_int32),
        ("cur_len", C.c_int32), ("prompt_len", C.c_int32), ("eos_id", C.c_int32),
        ("length_penalty", C.c_float),
    ]


NS_GEMM_GELU, NS_GEMM_DGELU, NS_GEMM_TN, NS_GEMM_ATOMIC32, NS_GEMM_DROP_A = 1, 2, 4, 8, 16
NS_GEMM_GELU_SAVE_GRAD, NS_GEMM_MUL_P16, NS_GEMM_COLSUM_A = 32, 64, 128

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol the header declares is listed here
# and checked at load time (tests/test_abi.py also cross-checks the header text).
SIGNATURES = {
    "ns_version": (C.c_int, []),
    "ns_last_error": (C.c_char_p, []),
    "ns_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "ns_debug_set_ring": (None, [C.c_int]),
    "ns_debug_set_ad_self": (None, [C.c_int]),
    "ns_gemm_ln_supported": (C.c_int, [_i, _i, _i, _i]),
    "ns_gemm_ln": (C.c_int, [C.POINTER(GemmLnDesc), _vp]),
    "ns_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "ns_layernorm_bwd": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ns_signal_pack": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ns_feed_pack": (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ns_embed_pos": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ns_dgelu_mul": (C.c_int, [_vp, _vp, _vp, C.POINTER(RowMap), _i, _i, _i, _vp]),
    "ns_colsum": (C.c_int, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "ns_cast_jobs": (C.c_int, [_vp, _i, _vp]),
    "ns_zero_spans": (C.c_int, [C.POINTER(Span), _i, _vp]),
    "ns_add_i32": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "ns_adalora_fold_grads": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "ns_adalora_fold_jobs": (C.c_int, [_vp, _i, _vp]),
    "ns_gemm_side_supported": (C.c_int, [_i, _i, _i]),
    "ns_gemm_side_reduce": (C.c_int, [_vp, _i, _i, _f, _vp, _i, _vp]),
    "ns_orth_reg": (C.c_int, [_vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "ns_orth_reg_workspace_bytes": (_sz, [_i]),
    "ns_lora_bwd_supported": (C.c_int, [_i, _i, _i]),
    "ns_lora_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ns_lora_bwd_dudb": (C.c_int, [C.POINTER(LoraBwdDesc), _vp]),
    "ns_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), _vp]),
    "ns_attn_bwd": (C.c_int, [C.POINTER(AttnDesc), _vp]),
    "ns_attn_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ns_cross_entropy": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ns_argmax_rows": (C.c_int, [_vp, _i, _i, _i, _vp, _vp]),
    "ns_attn_decode": (C.c_int, [C.POINTER(AttnDecodeDesc), _vp]),
    "ns_attn_fewq": (C.c_int, [C.POINTER(AttnFewqDesc), _vp]),
    "ns_vt_pack": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "ns_logits_process": (C.c_int, [C.POINTER(LogitsProcDesc), _vp]),
    "ns_topk_workspace_bytes": (C.c_size_t, [_i, C.c_longlong, _i]),
    "ns_topk_groups": (C.c_int, [_vp, _i, C.c_longlong, _i, _vp, _vp, _vp, _vp]),
    "ns_beam_update": (C.c_int, [C.POINTER(BeamDesc), _vp]),
    "ns_anc_update": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ns_greedy_update": (C.c_int, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ns_logits_select": (C.c_int, [C.POINTER(LogitsProcDesc), _i, _i, _vp, _vp, _vp]),
    "ns_topk_merge": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ns_grad_norm_workspace_bytes": (C.c_size_t, []),
    "ns_grad_norm": (C.c_int, [_vp, _sz, _vp, _vp, _vp, _vp]),
    "ns_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _sz, C.POINTER(AdamWCfg), _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared object; raise loudly when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NeuSpeechHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the product path)")
    # torch first: the library links libamdhip64 by soname, and the process must end up with ONE HIP runtime -- the one
    # torch brought (its streams / allocations are what the kernels are launched on).  Loaded the other way round, the
    # library binds /opt/rocm's runtime and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise NeuSpeechHipError(f"libneuspeech_hip.so lacks symbol {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ns_version() != ABI_VERSION:
        raise NeuSpeechHipError(f"ABI version mismatch: {lib.ns_version()}")
    if os.environ.get("NS_GEMM_MODE"):   # same-box A/B runs of the GEMM kernel variants (ns_debug_set_ring: 1 = auto, 4 = one tile per workgroup, ...)
        lib.ns_debug_set_ring(int(os.environ["NS_GEMM_MODE"]))
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ns_last_error().decode("utf-8", "replace")
        raise NeuSpeechHipError(f"{what} failed (status {rc}): {msg}")
