"""Thin torch-tensor front end over the C ABI (ctypes).  PyTorch supplies device
memory and streams only; every FLOP happens in libneuspeech_hip.so."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L
from .lib import (AdamWCfg, AttnDesc, CastJob, GemmDesc, RowMap, NS_GEMM_ATOMIC32, NS_GEMM_COLSUM_A, NS_GEMM_DGELU, NS_GEMM_DROP_A,
                  NS_GEMM_GELU, NS_GEMM_GELU_SAVE_GRAD, NS_GEMM_MUL_P16, NS_GEMM_TN)

__all__ = ["gemm", "rowmap", "ptr", "layernorm_fwd", "layernorm_bwd", "signal_pack", "feed_pack", "embed_pos", "attn_fwd",
           "attn_bwd", "cross_entropy", "dgelu_mul", "colsum", "argmax_rows", "grad_norm", "adamw_step", "cast_jobs", "make_cast_jobs",
           "NS_GEMM_GELU", "NS_GEMM_DGELU", "NS_GEMM_TN", "NS_GEMM_ATOMIC32", "NS_GEMM_DROP_A", "NS_GEMM_GELU_SAVE_GRAD",
           "NS_GEMM_MUL_P16", "NS_GEMM_COLSUM_A"]


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of torch's CURRENT stream on the current device (graph capture and the DP side stream switch it).
    The raw accessor is ~20x cheaper than torch.cuda.current_stream(): 374 launches per training step call this."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# ---- launch lists: a host-side stand-in for a hipGraph where capturing one does not pay (generate.py: short generations).
# While a list is being RECORDED on this thread, entry points are not launched but appended -- function, marshalled arguments -- exactly
# as a stream capture records instead of executing; replay() then issues them on the current stream at ~1 us of host time per launch,
# where building a descriptor through ctypes costs ~10 us (40 field stores, a dozen data_ptr() calls): a decode step of ~80 launches is
# 0.8-1.0 ms of host work built afresh against 0.84 ms of GPU time, i.e. host-bound on a slow host, and 0.1 ms replayed.
import threading as _threading

_tls = _threading.local()


class LaunchList:
    """Entries hold RAW device addresses (the marshalled ctypes arguments), exactly as a captured hipGraph does: `keep` is where the
    recorder parks whatever owns that memory (generate.py: the decode session whose tensors the launches point into), so a list can
    never outlive its buffers.  Only entry points that go through _call() are recorded: a torch op issued inside a recorded region runs
    once, at record time, and is absent from every replay -- NS_LISTS_VERIFY=1 makes generate() check a freshly recorded list against
    eager launches (tests/test_generate_gpu.py)."""
    __slots__ = ("calls", "keep")

    def __init__(self):
        self.calls = []
        self.keep = None

    def replay(self):
        st = _stream()
        for fn, name, args in self.calls:
            rc = fn(*args, st)
            if rc:
                L.check(rc, name)


class recording:
    """with ops.recording(launch_list): ... -- this thread's entry-point calls are recorded into the list, not launched"""

    def __init__(self, lst: LaunchList):
        self.lst = lst

    def __enter__(self):
        assert getattr(_tls, "rec", None) is None, "nested ops.recording"
        _tls.rec = self.lst
        return self.lst

    def __exit__(self, *exc):
        _tls.rec = None
        return False


STEP_PROFILE = None   # set to a list by bench.py: every entry-point launch is then bracketed by HIP events on the launch stream and recorded as
#                       (class, flop, algorithmic bytes, e0, e1); the class / work come from the wrapper (_work) or default to the entry point's name


def _work(cls: str, flop: float = 0.0, nbytes: float = 0.0):
    """wrapper -> _call: the class and the ALGORITHMIC work (FLOP, HBM bytes: every operand once) of the launch that follows"""
    if STEP_PROFILE is not None:
        _tls.work = (cls, float(flop), float(nbytes))


def _call(name: str, *args):
    fn = getattr(L.load(), name)
    rec = getattr(_tls, "rec", None)
    if rec is not None:
        rec.calls.append((fn, name, args))
        return
    if STEP_PROFILE is None:
        L.check(fn(*args, _stream()), name)
        return
    work = getattr(_tls, "work", None) or (name[3:] if name.startswith("ns_") else name, 0.0, 0.0)
    _tls.work = None
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    L.check(fn(*args, _stream()), name)
    e1.record(st)
    STEP_PROFILE.append((*work, e0, e1))


def ptr(t, off: int = 0) -> int:
    """Device address of tensor element `off` (0 for None)."""
    if t is None:
        return 0
    if isinstance(t, int):
        return t
    if isinstance(t, tuple):
        t, off = t
    return t.data_ptr() + off * t.element_size()


def rowmap(ld: int, seg_rows: int = 0, seg_stride: int = 0) -> RowMap:
    return RowMap(int(seg_stride), int(seg_rows), int(ld))


_ZERO_MAP = RowMap(0, 0, 0)
GEMM_PROFILE = None   # set to a list by bench.py to time every GEMM launch with HIP events


def _fill_gemm_desc(d, *, A, am, K, B, ldb=None, bm=None, M, N, A2=None, am2=None, K2=0, B2=None, ldb2=0, a2_ngroup=0,
                    bias=None, C16=None, c16m=None, G16=None, g16m=None, P16=None, p16m=None, R32=None, H32=None, h32m=None,
                    pos=None, pos_rows=0, C32=None, ldc32=0, flags=0, splits=1, drop_p=0.0, drop_seed=0, alpha=0.0,
                    side_B=None, side_ldb=0, side_n=0, side_out=None, side_drop_p=0.0, side_drop_seed=0, k_alg=None, seed_dev=None):
    d.side_B, d.side_ldb, d.side_n, d.side_out = ptr(side_B), side_ldb, side_n, ptr(side_out)
    d.side_drop_p, d.side_drop_seed = side_drop_p, side_drop_seed
    d.seed_dev = ptr(seed_dev)
    d.A, d.am, d.K = ptr(A), am, K
    d.B = ptr(B)
    d.bm = bm if bm is not None else rowmap(ldb)
    d.A2, d.am2, d.K2 = ptr(A2), (am2 if am2 is not None else _ZERO_MAP), K2
    d.B2, d.ldb2, d.a2_ngroup = ptr(B2), ldb2, a2_ngroup
    d.M, d.N = M, N
    d.bias = ptr(bias)
    d.C16, d.c16m = ptr(C16), (c16m if c16m is not None else _ZERO_MAP)
    d.G16, d.g16m = ptr(G16), (g16m if g16m is not None else _ZERO_MAP)
    d.P16, d.p16m = ptr(P16), (p16m if p16m is not None else _ZERO_MAP)
    d.R32, d.H32, d.h32m = ptr(R32), ptr(H32), (h32m if h32m is not None else _ZERO_MAP)
    d.pos, d.pos_rows = ptr(pos), pos_rows
    d.C32, d.ldc32 = ptr(C32), ldc32
    d.flags, d.splits = flags, splits
    d.drop_p, d.drop_seed, d.alpha = drop_p, drop_seed, alpha
    return d


def _gemm_kind(kw) -> str:
    """which kernel family ns_gemm dispatches this launch to (mirrors csrc/ns_gemm.hip)"""
    M, N, flags = kw["M"], kw["N"], kw.get("flags", 0)
    if flags & NS_GEMM_TN:
        return "tn"
    if N <= 96 or (M <= 1024 and N <= 4096):
        return "nt32"
    return "nt256" if (N >= 256 and (M >= 2048 or (M >= 128 and N >= 8192)) and ((M + 255) // 256) * ((N + 255) // 256) >= 192) else "nt128"


def _gemm_work(kw):
    """(FLOP, algorithmic HBM bytes) of one ns_gemm launch: every operand and every output once.  A row-mapped A (the conv stem's overlapping
    k = 3 windows over a halo image) counts the image once, not its three views."""
    M, N, K, K2, flags = kw["M"], kw["N"], kw["K"], kw.get("K2", 0), kw.get("flags", 0)
    side = kw.get("side_n", 0) if kw.get("side_B") is not None else 0      # side product: 2 M N side_n more
    flop = 2.0 * M * N * ((kw.get("k_alg") or K) + K2 + side)
    am = kw.get("am")
    if flags & NS_GEMM_TN:      # C32 (M x N) += A^T B over K reduction rows: A is (K, M), B is (K, N)
        by = 2.0 * K * (M + N) + 4.0 * M * N
        return flop, by
    if am is not None and am.seg_rows > 0 and am.ld < K:      # overlapping rows: the (segments x seg_stride) image once
        a_bytes = 2.0 * (M / am.seg_rows) * am.seg_stride
    else:
        a_bytes = 2.0 * M * K
    by = a_bytes + 2.0 * N * K + 2.0 * (M + N) * K2
    for name, w in (("C16", 2), ("G16", 2), ("P16", 2), ("R32", 4), ("H32", 4), ("C32", 4)):
        if kw.get(name) is not None:
            by += float(w) * M * N
    if kw.get("bias") is not None:
        by += 4.0 * N
    if side:
        by += 2.0 * side * N + 4.0 * ((N + 255) // 256) * M * side
    return flop, by


def _gemm_epi(kw) -> str:
    flags = kw.get("flags", 0)
    e = "res" if kw.get("H32") is not None else ("dgelu" if flags & (NS_GEMM_DGELU | NS_GEMM_MUL_P16) else ("gelu" if flags & NS_GEMM_GELU else "plain"))
    if kw.get("side_B") is not None:
        e += "+side"
    if kw.get("K2", 0):
        e += "+k2"
    if kw.get("drop_p", 0.0) > 0.0:
        e += "+drop"
    return e


def gemm(**kw):
    """ns_gemm (keyword arguments = the descriptor's fields, see _fill_gemm_desc).  k_alg: the ALGORITHMIC reduction length where K
    carries padding (the first conv: 3 x ch against 3 x ch_pad); only the bench's FLOP count reads it"""
    d = _fill_gemm_desc(GemmDesc(), **kw)
    if STEP_PROFILE is not None:
        kind = _gemm_kind(kw)
        _work(f"gemm {kind} {_gemm_epi(kw)} K={kw['K']} N={kw['N']}" if kind == "nt256" else f"gemm {kind}", *_gemm_work(kw))
    if GEMM_PROFILE is None:
        _call("ns_gemm", C.byref(d))
        return
    # bench.py roofline leg: HIP events on the launch stream around every GEMM launch
    kind = _gemm_kind(kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    _call("ns_gemm", C.byref(d))
    e1.record(torch.cuda.current_stream())
    fl, by = _gemm_work(kw)
    GEMM_PROFILE.append((kind, fl, e0, e1, by))


def gemm_ln_supported(M: int, N: int, K: int, K2: int = 0, lda: int | None = None) -> bool:
    """the shape test of the library AND ns_gemm_ln's operand-extent test (32-bit buffer offsets: the last row of A, row stride lda, must
    end below 2 GiB -- e.g. fc2 of a very large batch), so that a caller falls back to ns_gemm + ns_layernorm_fwd instead of raising"""
    lda = K if lda is None else lda
    return bool(L.load().ns_gemm_ln_supported(M, N, K, K2)) and 2 * ((M - 1) * lda + K + 64) < 0x7FFF0000


def gemm_ln(*, gamma, beta, x16, ldx, mean=None, rstd=None, eps=1e-5, **kw):
    """ns_gemm_ln: the residual Linear described by **kw (as for gemm: N = 512, R32, H32) + the LayerNorm that reads H32"""
    q = L.GemmLnDesc()
    _fill_gemm_desc(q.g, **kw)
    q.gamma, q.beta, q.eps, q.ldx = ptr(gamma), ptr(beta), eps, ldx
    q.x16, q.mean, q.rstd = ptr(x16), ptr(mean), ptr(rstd)
    fl, by = _gemm_work(kw)
    by += kw["M"] * (2.0 * kw["N"] + 8.0)      # + the LayerNorm's x16 row and its statistics
    _work("gemm rowln (+LayerNorm)", fl, by)
    if GEMM_PROFILE is None:
        _call("ns_gemm_ln", C.byref(q))
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    _call("ns_gemm_ln", C.byref(q))
    e1.record(torch.cuda.current_stream())
    GEMM_PROFILE.append(("rowln", fl, e0, e1, by))


def gemm_side_supported(M: int, N: int, K: int) -> bool:
    return bool(L.load().ns_gemm_side_supported(M, N, K))


def gemm_side_reduce(slabs, tiles, M, alpha, u16, ldu):
    _work("side_reduce", 0.0, M * 32 * (4.0 * tiles + 2.0))
    _call("ns_gemm_side_reduce", ptr(slabs), tiles, M, alpha, ptr(u16), ldu)


def layernorm_fwd(x32, gamma, beta, y16, mean, rstd, rows, d, y32=None, eps=1e-5):
    _work("layernorm_fwd", 0.0, rows * (d * (6.0 + (4.0 if y32 is not None else 0.0)) + 8.0))
    _call("ns_layernorm_fwd", ptr(x32), ptr(gamma), ptr(beta), ptr(y16), ptr(y32), ptr(mean), ptr(rstd),
                                      rows, d, eps)


def layernorm_bwd(dy, dy_is_f32, x32, mean, rstd, gamma, dres, dx32, dx16, rows, d):
    _work("layernorm_bwd", 0.0, rows * (d * ((4.0 if dy_is_f32 else 2.0) + 4.0 + (4.0 if dres is not None else 0.0)
                                              + (4.0 if dx32 is not None else 0.0) + (2.0 if dx16 is not None else 0.0)) + 8.0))
    _call("ns_layernorm_bwd", ptr(dy), int(dy_is_f32), ptr(x32), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres),
                                      ptr(dx32), ptr(dx16), rows, d)


def signal_pack(x32, out16, B, ch, T, Cp):
    _work("signal_pack", 0.0, B * (4.0 * ch * T + 2.0 * (T + 2) * Cp))
    _call("ns_signal_pack", ptr(x32), ptr(out16), B, ch, T, Cp)


def feed_pack(items_dev, B, ch, T, Cp, out16, x32=None):
    """items_dev: device uint8/int64 tensor holding B ns_feed_item records (32 bytes each)"""
    _call("ns_feed_pack", ptr(items_dev), B, ch, T, Cp, ptr(out16), ptr(x32))


def embed_pos(ids, E32, P32, h32, rows, Lseq, d, pos0=0, pos0_dev=None):
    _call("ns_embed_pos", ptr(ids), ptr(E32), ptr(P32), ptr(h32), rows, Lseq, d, pos0, ptr(pos0_dev))


def dgelu_mul(a16, pre16, out16, out_map, rows, cols, pre_is_grad=False):
    _call("ns_dgelu_mul", ptr(a16), ptr(pre16), ptr(out16), C.byref(out_map), rows, cols, int(pre_is_grad))


def colsum(a16, out32, rows, cols, ld, alpha=1.0):
    _call("ns_colsum", ptr(a16), ptr(out32), rows, cols, ld, alpha)


def _attn_desc(Q, K, V, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, causal, LSE=None, dO=None, dQ=None, dK=None, dV=None,
               Delta=None, lddo=0, lddq=0, lddk=0, lddv=0, workspace=None):
    d = AttnDesc()
    if workspace is not None:       # one-pass backward (ns_attn_bwd_workspace_bytes > 0 for this shape)
        d.workspace, d.workspace_bytes = ptr(workspace), workspace.numel() * workspace.element_size()
    d.Q, d.K, d.V, d.O = ptr(Q), ptr(K), ptr(V), ptr(O)
    d.dO, d.dQ, d.dK, d.dV = ptr(dO), ptr(dQ), ptr(dK), ptr(dV)
    d.LSE, d.Delta = ptr(LSE), ptr(Delta)
    d.B, d.H, d.Lq, d.Lk, d.head_dim = B, H, Lq, Lk, 64
    d.ldq, d.ldk, d.ldv, d.ldo = ldq, ldk, ldv, ldo
    d.lddo, d.lddq, d.lddk, d.lddv = lddo, lddq, lddk, lddv
    d.causal = int(causal)
    return d


def _attn_work(kw, bwd: bool):
    B, H, Lq, Lk = kw["B"], kw["H"], kw["Lq"], kw["Lk"]
    causal = bool(kw.get("causal"))
    fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
    by = 2.0 * B * H * 64 * (2 * Lq + 2 * Lk) + 4.0 * B * H * Lq         # Q, O, K, V + LSE
    if bwd:
        return 2.5 * fl, 2.0 * by + 4.0 * B * H * Lq                       # + dO, dQ, dK, dV, delta (five products against two)
    return fl, by


def attn_fwd(**kw):
    d = _attn_desc(**kw)
    _work("attn_fwd" + (" causal" if kw.get("causal") else "") + (" (few queries)" if kw["Lq"] <= 64 else ""), *_attn_work(kw, False))
    _call("ns_attn_fwd", C.byref(d))


def attn_bwd_workspace_bytes(B, H, Lq, Lk, causal=False) -> int:
    return int(L.load().ns_attn_bwd_workspace_bytes(B, H, Lq, Lk, int(causal)))


def attn_bwd(**kw):
    d = _attn_desc(**kw)
    _work("attn_bwd" + (" one pass" if kw.get("workspace") is not None and kw["Lq"] > 64 else "") + (" causal" if kw.get("causal") else "")
          + (" (few queries)" if kw["Lq"] <= 64 else ""), *_attn_work(kw, True))
    _call("ns_attn_bwd", C.byref(d))


def cross_entropy(logits16, labels, rows, V, ldv, row_loss, dlogits16, nvalid_dev, loss_scale_dev, loss_dev):
    _work("cross_entropy", 0.0, rows * V * (2.0 + (2.0 if dlogits16 is not None else 0.0)))
    _call("ns_cross_entropy", ptr(logits16), ptr(labels), rows, V, ldv, ptr(row_loss), ptr(dlogits16),
                                      ptr(nvalid_dev), ptr(loss_scale_dev), ptr(loss_dev))


def argmax_rows(logits16, rows, V, ldv, out):
    _call("ns_argmax_rows", ptr(logits16), rows, V, ldv, ptr(out))


def grad_norm(g32, n, workspace, norm2_dev, found_inf_dev):
    _work("grad_norm", 0.0, 4.0 * n)
    _call("ns_grad_norm", ptr(g32), n, ptr(workspace), ptr(norm2_dev), ptr(found_inf_dev))


def adamw_step(p, g, m, v, n, cfg: AdamWCfg, step_dev, norm2_dev, found_inf_dev, loss_scale_dev, growth_tracker_dev):
    _work("adamw_step", 0.0, 28.0 * n)
    _call("ns_adamw_step", ptr(p), ptr(g), ptr(m), ptr(v), n, C.byref(cfg), ptr(step_dev), ptr(norm2_dev),
                                   ptr(found_inf_dev), ptr(loss_scale_dev), ptr(growth_tracker_dev))


def zero_(*tensors):
    """clear up to 8 tensors per launch with ns_zero_spans (a kernel, never a memset node: see the header)"""
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    for i in range(0, len(ts), 8):
        chunk = ts[i:i + 8]
        arr = (L.Span * len(chunk))()
        for j, t in enumerate(chunk):
            assert t.is_contiguous()
            arr[j] = L.Span(t.data_ptr(), t.numel() * t.element_size())
        _work("zero_spans", 0.0, float(sum(t.numel() * t.element_size() for t in chunk)))
        _call("ns_zero_spans", arr, len(chunk))


def zeros(*shape, device, dtype):
    """torch.zeros without torch's fill kernel"""
    t = torch.empty(*shape, device=device, dtype=dtype)
    zero_(t)
    return t


def add_i32(counter_dev, v=1, n=1):
    """v added to n consecutive int32 device counters in one launch"""
    _call("ns_add_i32", ptr(counter_dev), int(n), int(v))


def make_cast_jobs(jobs, device) -> tuple[torch.Tensor, int]:
    """jobs: list of (src_ptr, dst_ptr, rows, cols, ld_src, ld_dst, scale, transpose) -> device table."""
    arr = (CastJob * len(jobs))()
    for i, job in enumerate(jobs):
        s, dptr, r, c, lds, ldd, sc, tr = job[:8]
        arr[i] = CastJob(s, dptr, r, c, lds, ldd, sc, int(tr), job[8] if len(job) > 8 else 0)
    raw = bytes(arr)
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    return t, len(jobs)


def cast_jobs(table: torch.Tensor, njobs: int):
    _call("ns_cast_jobs", ptr(table), njobs)


# ----------------------------------------------------------------------------- decode loop
def attn_decode(*, Q, K, V, O, groups, nq, H, Lk, Lk_max, ldq, ldk, ldv, ldo, anc=None, anc_ld=0, kv_group_stride=0,
                kv_pos_stride=0, kv_len_dev=None, Knew=None, Vnew=None, ldnew=0, slot0=0):
    d = L.AttnDecodeDesc()
    d.Q, d.K, d.V, d.O = ptr(Q), ptr(K), ptr(V), ptr(O)
    d.anc, d.kv_len_dev = ptr(anc), ptr(kv_len_dev)
    d.groups, d.nq, d.H, d.Lk, d.Lk_max, d.head_dim = groups, nq, H, Lk, Lk_max, 64
    d.ldq, d.ldk, d.ldv, d.ldo, d.anc_ld = ldq, ldk, ldv, ldo, anc_ld
    d.kv_group_stride, d.kv_pos_stride = kv_group_stride, kv_pos_stride
    d.Knew, d.Vnew, d.ldnew, d.slot0 = ptr(Knew), ptr(Vnew), ldnew, slot0
    _call("ns_attn_decode", C.byref(d))


def attn_fewq(*, Q, K, Vt, O, groups, nq, H, Lk, ldq, ldk, ldvt, ldo):
    d = L.AttnFewqDesc()
    d.Q, d.K, d.Vt, d.O = ptr(Q), ptr(K), ptr(Vt), ptr(O)
    d.groups, d.nq, d.H, d.Lk, d.ldq, d.ldk, d.ldvt, d.ldo = groups, nq, H, Lk, ldq, ldk, ldvt, ldo
    _call("ns_attn_fewq", C.byref(d))


def vt_pack(v16, ldv, vt16, groups, H, Lk, ldvt):
    _call("ns_vt_pack", ptr(v16), ldv, ptr(vt16), groups, H, Lk, ldvt)


def logits_process(*, logits16, scores32, ids, rows, V, ldv, ids_ld, cur_len, begin_index, log_softmax, beam_scores=None,
                   repetition_penalty=1.0, no_repeat_ngram=0, suppress=None, n_suppress=0, begin_suppress=None,
                   n_begin_suppress=0, cur_len_dev=None, bias1=None, seq_tok=None, seq_off=None, seq_bias=None, n_seq=0,
                   forced=None, n_forced=0):
    d = L.LogitsProcDesc()
    d.forced, d.n_forced = ptr(forced), n_forced
    d.logits16, d.scores32, d.ids, d.beam_scores = ptr(logits16), ptr(scores32), ptr(ids), ptr(beam_scores)
    d.suppress, d.begin_suppress, d.cur_len_dev = ptr(suppress), ptr(begin_suppress), ptr(cur_len_dev)
    d.rows, d.V, d.ldv, d.ids_ld, d.cur_len, d.begin_index = rows, V, ldv, ids_ld, cur_len, begin_index
    d.n_suppress, d.n_begin_suppress, d.no_repeat_ngram, d.log_softmax = n_suppress, n_begin_suppress, no_repeat_ngram, int(log_softmax)
    d.repetition_penalty = repetition_penalty
    d.bias1, d.seq_tok, d.seq_off, d.seq_bias, d.n_seq = ptr(bias1), ptr(seq_tok), ptr(seq_off), ptr(seq_bias), n_seq
    _call("ns_logits_process", C.byref(d))


SELECT_MAX_LDV = 26 * 256 * 8


def logits_select(*, logits16, ids, rows, V, ldv, ids_ld, cur_len, begin_index, log_softmax, k, group_rows, cand_vals,
                  cand_idx, beam_scores=None, repetition_penalty=1.0, no_repeat_ngram=0, suppress=None, n_suppress=0,
                  begin_suppress=None, n_begin_suppress=0, cur_len_dev=None, scores32=None, forced=None, n_forced=0,
                  bias1=None, seq_tok=None, seq_off=None, seq_bias=None, n_seq=0):
    d = L.LogitsProcDesc()
    d.forced, d.n_forced = ptr(forced), n_forced
    d.bias1, d.seq_tok, d.seq_off, d.seq_bias, d.n_seq = ptr(bias1), ptr(seq_tok), ptr(seq_off), ptr(seq_bias), n_seq
    d.logits16, d.scores32, d.ids, d.beam_scores = ptr(logits16), None, ptr(ids), ptr(beam_scores)
    d.suppress, d.begin_suppress, d.cur_len_dev = ptr(suppress), ptr(begin_suppress), ptr(cur_len_dev)
    d.rows, d.V, d.ldv, d.ids_ld, d.cur_len, d.begin_index = rows, V, ldv, ids_ld, cur_len, begin_index
    d.n_suppress, d.n_begin_suppress, d.no_repeat_ngram, d.log_softmax = n_suppress, n_begin_suppress, no_repeat_ngram, int(log_softmax)
    d.repetition_penalty = repetition_penalty
    _call("ns_logits_select", C.byref(d), k, group_rows, ptr(cand_vals), ptr(cand_idx))


def topk_merge(cand_vals, cand_idx, groups, ncand, k, vals, idx):
    _call("ns_topk_merge", ptr(cand_vals), ptr(cand_idx), groups, ncand, k, ptr(vals), ptr(idx))


_topk_ws = {}


def topk_groups(x32, groups, n, k, vals, idx):
    key = (x32.device, groups, n, k)
    ws = _topk_ws.get(key)
    if ws is None:
        ws = torch.empty(L.load().ns_topk_workspace_bytes(groups, n, k), device=x32.device, dtype=torch.uint8)
        _topk_ws[key] = ws
    _call("ns_topk_groups", ptr(x32), groups, n, k, ptr(vals), ptr(idx), ptr(ws))


def beam_update(**kw):
    d = L.BeamDesc()
    for k, v in kw.items():
        setattr(d, k, ptr(v) if isinstance(v, (torch.Tensor, tuple)) or v is None else v)
    _call("ns_beam_update", C.byref(d))


def anc_update(anc_in, anc_out, parent, rows, ld, cur, cur_dev=None):
    _call("ns_anc_update", ptr(anc_in), ptr(anc_out), ptr(parent), rows, ld, cur, ptr(cur_dev))


def greedy_update(scores, rows, V, seqs, ld, cur, eos, pad, done, any_open, next_tok, cur_dev=None, best_idx=None):
    _call("ns_greedy_update", ptr(scores), rows, V, ptr(seqs), ld, cur, ptr(cur_dev), eos, pad, ptr(done), ptr(any_open),
                                      ptr(next_tok), ptr(best_idx))


# ----------------------------------------------------------------------------- AdaLoRA
def adalora_fold_grads(dBf, B, E, dB, dE, N, r, s):
    _call("ns_adalora_fold_grads", ptr(dBf), ptr(B), ptr(E), ptr(dB), ptr(dE), N, r, s)


def make_fold_jobs(jobs, device):
    """jobs: list of (dBf_ptr, B_ptr, E_ptr, dB_ptr, dE_ptr, N, r, s)"""
    arr = (L.AdaloraFoldJob * len(jobs))()
    for i, j in enumerate(jobs):
        arr[i] = L.AdaloraFoldJob(*j)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(jobs)


def adalora_fold_jobs(table, njobs):
    _call("ns_adalora_fold_jobs", ptr(table), njobs)


def make_orth_jobs(jobs, device):
    """jobs: list of (P_ptr, G_ptr, r, len, ld, is_b)"""
    arr = (L.OrthJob * len(jobs))()
    for i, j in enumerate(jobs):
        arr[i] = L.OrthJob(*j)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(jobs)


_orth_ws = {}
_retired_ws = []   # outgrown workspaces stay allocated: graphs captured while they were current replay with their address


def orth_reg(table, njobs, weight_over_num, loss_scale_dev, reg_out_dev):
    need = L.load().ns_orth_reg_workspace_bytes(njobs)
    ws = _orth_ws.get(reg_out_dev.device)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _retired_ws.append(ws)      # a captured training step (hipGraph) may still hold its address
        ws = torch.empty(need, device=reg_out_dev.device, dtype=torch.uint8)
        _orth_ws[reg_out_dev.device] = ws
    _call("ns_orth_reg", ptr(table), njobs, weight_over_num, ptr(loss_scale_dev), ptr(reg_out_dev), ptr(ws), ws.numel())


# ----------------------------------------------------------------------------- LoRA backward (du + dB in one pass over dy)
def lora_bwd_supported(N: int, r: int, G: int) -> bool:
    return bool(L.load().ns_lora_bwd_supported(N, r, G))


_lora_ws = {}


def lora_bwd_dudb(*, dy, ldy, u, ldu, du, lddu, sBT, dB, lddb, M, N, r, alpha_du, alpha_db, splits=0, slabs=True):
    """sBT / dB / alpha_db: one entry per column group of dy (q | k | v: three, else one).  slabs: dB partials through a
    workspace + reduce launch (one cached workspace per device, sized for the largest request) instead of fp32 atomics"""
    d = L.LoraBwdDesc()
    dev = (du[0] if isinstance(du, tuple) else du).device      # (tensor, element offset) pairs address a column group of a stacked buffer
    if slabs:
        need = L.load().ns_lora_bwd_workspace_bytes(M, N, len(sBT), splits)
        ws = _lora_ws.get(dev)
        if ws is None or ws.numel() < need:
            if ws is not None:
                _retired_ws.append(ws)
            ws = torch.empty(need, device=dev, dtype=torch.uint8)
            _lora_ws[dev] = ws
        d.workspace, d.workspace_bytes = ptr(ws), ws.numel()
    d.dy, d.u, d.du = ptr(dy), ptr(u), ptr(du)
    G = len(sBT)
    for g in range(G):
        d.sBT[g], d.dB[g], d.alpha_db[g] = ptr(sBT[g]), ptr(dB[g]), alpha_db[g]
    d.M, d.N, d.r, d.G = M, N, r, G
    d.ldy, d.ldu, d.lddu, d.lddb = ldy, ldu, lddu, lddb
    d.alpha_du, d.splits = alpha_du, splits
    _work("lora_bwd_dudb (+reduce)", 2.0 * M * G * N * r * 2, 2.0 * M * G * N + 2.0 * M * G * r * 2 + 2.0 * G * N * r + 4.0 * G * N * r)
    _call("ns_lora_bwd_dudb", C.byref(d))
