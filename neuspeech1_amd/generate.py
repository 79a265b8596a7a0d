"""Greedy / beam-search generation on top of the engine's weights (evaluation.py:369-386 call shape).

Design notes (MI355X-first):
  * the encoder runs once; each decoder layer's cross-attention K/V is projected once per SEQUENCE and is
    never duplicated per beam nor re-gathered: ns_attn_decode reads it once for all beams of the sequence;
  * the self-attention cache is laid out [position][slot]; beams are reordered by rewriting a small int32
    ancestry table (which slot holds my token-p K/V) instead of index_select-ing every cache tensor;
  * logits processors, top-2k selection and the HF beam bookkeeping run on the device; the host only polls a
    two-word flag every `check_every` steps;
  * a decode iteration is ~160 small launches, so after one eager iteration the two halves of an iteration
    ("select the next token", "feed it through the decoder") are captured as hipGraphs -- one pair per ping-pong
    parity -- and replayed; the position / length the kernels need comes from device counters the graph itself
    advances (the C ABI's *_dev arguments), never from launch arguments.
"""
from __future__ import annotations

import collections
import os
import time

import numpy as np
import torch

from . import ops
from .lib import GPU_CAPTURE_LOCK
from .ops import rowmap

F16, F32 = torch.float16, torch.float32


def _sequence_bias_tables(sequence_bias, V, dev):
    """{(token ids): bias} or [[[token ids], bias], ...] (both forms of HF's `sequence_bias`) -> device tables of
    ns_logits_proc_desc: a dense vector for the length-1 sequences, flattened tokens + offsets + biases for the rest.
    Validation as HF generation/logits_process.py (SequenceBiasLogitsProcessor._validate_arguments)."""
    if sequence_bias is None:
        return {}
    if isinstance(sequence_bias, list):
        if len(sequence_bias) == 0 or any(len(s_) != 2 or not isinstance(s_[0], (list, tuple)) or len(s_[0]) == 0 for s_ in sequence_bias):
            raise ValueError(f"`sequence_bias` has to be a non-empty dictionary, or non-empty list of lists but is {sequence_bias}.")
        sequence_bias = {tuple(s_[0]): s_[1] for s_ in sequence_bias}
    if not isinstance(sequence_bias, dict) or len(sequence_bias) == 0:
        raise ValueError(f"`sequence_bias` has to be a non-empty dictionary, or non-empty list of lists but is {sequence_bias}.")
    for ids, bias in sequence_bias.items():
        if not isinstance(ids, tuple) or len(ids) == 0 or any((not isinstance(t, (int, np.integer))) or t < 0 for t in ids):
            raise ValueError(f"Each key in `sequence_bias` has to be a non-empty tuple of positive integers, but is {sequence_bias}.")
        if not isinstance(bias, float):
            raise ValueError(f"`sequence_bias` has to be a dict with floats as values, but is {sequence_bias}.")
    bad = [t for ids in sequence_bias for t in ids if t >= V]
    if bad:
        raise ValueError(f"The model vocabulary size is {V}, but the following tokens were being biased: {bad}")
    out = {}
    ones = {ids[0]: b for ids, b in sequence_bias.items() if len(ids) == 1}
    if ones:
        b1 = torch.zeros(V, dtype=F32)
        b1[list(ones.keys())] = torch.tensor(list(ones.values()), dtype=F32)
        out["bias1"] = b1.to(dev)
    longer = [(ids, b) for ids, b in sequence_bias.items() if len(ids) > 1]
    if longer:
        off = np.cumsum([0] + [len(ids) for ids, _ in longer]).astype(np.int32)
        out["seq_tok"] = torch.from_numpy(np.concatenate([np.asarray(ids, dtype=np.int32) for ids, _ in longer])).to(dev)
        out["seq_off"] = torch.from_numpy(off).to(dev)
        out["seq_bias"] = torch.tensor([b for _, b in longer], dtype=F32).to(dev)
        out["n_seq"] = len(longer)
    return out


class Generator:
    def __init__(self, engine, use_graph: bool = True, graph_min_steps: int | None = None, cross_mfma: bool = True):
        self.eng = engine
        self.use_graph = use_graph
        self.cross_mfma = cross_mfma    # beam cross-attention through ns_attn_fewq (False: the flash kernel)
        # capturing the four graphs costs ~3 ms; measured on MI355X the decode loop is GPU-bound (B = 128: a replayed step
        # and an eagerly launched one take the same time), so graphs only insure against a slow / contended host.  At 64
        # new tokens the capture is never paid back (tools/probe/gen_timing.py: 82.0 vs 78.9 ms greedy, 104.6 vs 101.7 ms
        # beam-5), so only generations with >= 128 steps left (whisper's max_length is 448) capture
        import os
        self.graph_min_steps = int(os.environ.get("NS_GRAPH_MIN_STEPS", 128)) if graph_min_steps is None else graph_min_steps
        self.use_lists = os.environ.get("NS_LAUNCH_LISTS", "1") != "0"      # recorded launch lists for generations too short for graphs
        self.adaptive = os.environ.get("NS_DECODE_ADAPT", "1") != "0"       # lists -> hipGraphs when the replays turn out host-bound
        self.adaptive_min_steps = int(os.environ.get("NS_DECODE_ADAPT_MIN_STEPS", 24))
        # share of a chunk's wall time spent inside the replays that counts as host-bound.  (0.8 fired on a fast host slowed by 8 us per
        # launch -- 0.72 ms of replays inside 0.86 ms of GPU time: still GPU-bound, and the capture cost 2.2 ms of a 77-ms generation:
        # tools/probe/decode_slow_host.py)
        self.adaptive_frac = 0.95
        self.last_loop_mode = None
        # decode SESSIONS: the device state of a generation (caches, score tables, token buffers, cross K/V) and the recorded launch
        # lists / captured hipGraphs of its loop, kept per call signature: an evaluation run calls generate() with the same shapes and
        # processors for every batch, and from the second call on nothing is allocated, recorded or captured again (the first call of
        # a signature replays launch lists, the second captures the graphs, every later one replays them from the first step)
        self.cache_sessions = os.environ.get("NS_DECODE_SESSIONS", "1") != "0"
        self.max_sessions = 2
        self.max_session_frac = 0.25      # of the device memory, all sessions together (large-v2 at B = 128, beam 5: 62 GB of cross K|V images each)
        self._sessions = collections.OrderedDict()
        self.capture_failures = 0           # hipGraph captures that were invalidated (one disables graphs for this Generator)
        self._graph_graveyard = []          # graph objects of a capture that did not end cleanly: never destroyed (see run_loop.build)

    def clear_sessions(self):
        """release the device state, launch lists and hipGraphs kept for the call signatures seen so far"""
        self._sessions.clear()

    def _weights_fingerprint(self):
        """device addresses of everything a recorded decoder launch points at: a reloaded / re-merged engine gets new sessions"""
        eng = self.eng
        ptrs = [eng.E16.data_ptr(), eng.E32.data_ptr(), eng.dec_pos.data_ptr()] + [t.data_ptr() for t in eng.dec_ln]
        for Lw in eng.dec:
            for v in Lw.values():
                if isinstance(v, tuple):
                    ptrs += [t.data_ptr() for t in v if torch.is_tensor(t)]
                elif hasattr(v, "w"):
                    ptrs += [v.w.data_ptr(), v.bias.data_ptr() if v.bias is not None else 0]
        return hash(tuple(ptrs))

    @torch.no_grad()
    def generate(self, x32: torch.Tensor, prompt: torch.Tensor, **kw) -> torch.Tensor:
        """see _generate.  NS_LISTS_VERIFY=1 (debug): whenever the loop replayed recorded launch lists, the same call is run
        again with eager launches only and the ids compared -- a torch op that slipped into a recorded region (it would run once, at
        record time, and be missing from every replay) or a list that outlived a buffer shows up as a difference here"""
        out = self._generate(x32, prompt, **kw)
        import os
        if os.environ.get("NS_LISTS_VERIFY") == "1" and str(self.last_loop_mode).startswith("lists") and kw.get("trace") is None:
            mode = self.last_loop_mode
            saved = (self.use_lists, self.use_graph, self.cache_sessions)
            self.use_lists = self.use_graph = self.cache_sessions = False
            try:
                ref = self._generate(x32, prompt, **kw)
            finally:
                self.use_lists, self.use_graph, self.cache_sessions = saved
            self.last_loop_mode = mode
            if ref.shape != out.shape:
                raise RuntimeError(f"NS_LISTS_VERIFY: replayed launch lists ({mode}) gave ids of shape {tuple(out.shape)}, eager launches {tuple(ref.shape)}")
            if not torch.equal(ref, out):
                bad = (ref != out).any(dim=1).nonzero().flatten().tolist()
                raise RuntimeError(f"NS_LISTS_VERIFY: replayed launch lists ({mode}) and eager launches disagree on rows {bad[:8]}")
        return out

    @torch.no_grad()
    def _generate(self, x32: torch.Tensor, prompt: torch.Tensor, num_beams: int = 1, max_new_tokens: int = 64,
                 repetition_penalty: float = 1.0, no_repeat_ngram_size: int = 0, suppress_tokens=(),
                 begin_suppress_tokens=(), length_penalty: float = 1.0, eos_id: int | None = None,
                 pad_id: int | None = None, check_every: int = 4, sequence_bias=None, forced_decoder_ids=None,
                 begin_index: int | None = None, trace: list | None = None) -> torch.Tensor:
        """forced_decoder_ids: [[position, token or None], ...] (generation_config.forced_decoder_ids as the reference's
        wrapper passes them on, utils/load_model.py:1210-1256): HF ForceTokensLogitsProcessor semantics, position =
        absolute index in the decoder sequence.  begin_index: the position the begin-suppress list applies at (default:
        the prompt length; HF of the reference's era adds forced_decoder_ids[-1][0]).  trace (tests only, greedy only): a list
        that receives, per selection step, (values, columns) of each row's two best PROCESSED scores -- this path's own
        decision margin, what tests/test_live_fp16_gpu.py holds a flipped token against; the loop then stays eager."""
        eng = self.eng
        if getattr(eng, "dec_lora", False):
            raise RuntimeError("decode from merged weights (merge_and_unload / merge_lora.py): the generation loop does not "
                               "carry decoder adapters")
        dims, dev = eng.dims, eng.dev
        d, H, S, V, Vp = dims.d, dims.heads, dims.src_pos, dims.vocab, dims.vocab_pad
        eos = dims.eos_id if eos_id is None else eos_id
        pad = dims.pad_id if pad_id is None else pad_id
        B, P = prompt.shape
        nb = num_beams
        Bp = B * nb
        max_len = min(P + max_new_tokens, dims.tgt_pos)
        fewq = self.cross_mfma and 1 < nb <= 16
        fused_select = dims.vocab_pad <= ops.SELECT_MAX_LDV and os.environ.get("NS_NO_FUSED_SELECT") != "1"
        # ---- the session of this call signature (everything a recorded / captured launch has baked into it)
        key = None
        if self.cache_sessions and trace is None and dev.type == "cuda":
            key = (B, P, nb, max_len, fewq, fused_select, float(repetition_penalty), int(no_repeat_ngram_size),
                   tuple(int(t) for t in suppress_tokens), tuple(int(t) for t in begin_suppress_tokens), float(length_penalty), eos, pad,
                   repr(forced_decoder_ids), begin_index, repr(sequence_bias), str(dev), torch.cuda.current_stream().cuda_stream,
                   self._weights_fingerprint())
        ws = self._sessions.get(key) if key is not None else None
        if ws is None:
            ws = {"t": {}, "graphs": None, "lists": None, "calls": 0}
            if key is not None:
                self._sessions[key] = ws
                while len(self._sessions) > self.max_sessions:
                    self._sessions.popitem(last=False)
        elif key is not None:
            self._sessions.move_to_end(key)
        first_call = ws["calls"] == 0

        def T(name, make, init=None):
            """a tensor of the session: made by the first call of the signature, re-initialised by `init` on every later one"""
            t = ws["t"].get(name)
            if t is None:
                t = ws["t"][name] = make()
            elif init is not None:
                init(t)
            return t
        # ---- encoder + per-sequence cross K/V
        eng.training_mode = False
        eng._cur_seed = 0
        b = eng._alloc(B, 0, False)
        eng._b = b
        enc16 = eng.encode(x32.contiguous(), b, False)
        M = B * S
        kvx, vtx = [], []
        Sp = (S + 31) // 32 * 32
        # beams of a sequence = the few query rows of ns_attn_fewq (75 k tokens/s at beam 5, B = 128, against 71 k with
        # the 128-row flash kernel and 64 k with the per-key VALU kernel); the single greedy row stays on
        # ns_attn_decode (same speed, no transposed copy)
        for li, Lw in enumerate(eng.dec):
            t = T(f"kvx{li}", lambda: torch.empty(M, 2 * d, device=dev, dtype=F16))
            eng._lin(enc16, M, Lw["ckv"], C16=t)
            kvx.append(t)
            if fewq:
                # the cross V of a sequence is written once and read at every step: keep it transposed so the value
                # product's MFMA operand is one 16-B load per lane (ns_attn_fewq)
                vt = T(f"vtx{li}", lambda: ops.zeros(B, H, 64, Sp, device=dev, dtype=F16))    # (columns [S, Sp) stay zero)
                ops.vt_pack((t, d), 2 * d, vt, B, H, S, Sp)
                vtx.append(vt)
        # ---- decode state
        nl = dims.dec_layers
        # (a reused cache needs no clearing: a step reads positions below its own length only, all of them written by this generation;
        # the ancestry table of step t is rebuilt from its first t entries)
        kvc = [T(f"kvc{i}", lambda: torch.zeros(max_len * Bp, 2 * d, device=dev, dtype=F16)) for i in range(nl)]
        anc = [T(f"anc{i}", lambda: torch.zeros(Bp, max_len, device=dev, dtype=torch.int32)) for i in range(2)]
        if nb == 1 and first_call:     # greedy rows never change slots: the ancestry table is the identity, written once (no ns_anc_update per step)
            for a_ in anc:
                a_.copy_(torch.arange(Bp, device=dev, dtype=torch.int32).unsqueeze(1).expand(Bp, max_len))
        h = [T(f"h{i}", lambda: torch.empty(Bp, d, device=dev, dtype=F32)) for i in range(2)]
        x16 = T("x16", lambda: torch.empty(Bp, d, device=dev, dtype=F16))
        qkv = T("qkv", lambda: torch.empty(Bp, 3 * d, device=dev, dtype=F16))
        qc = T("qc", lambda: torch.empty(Bp, d, device=dev, dtype=F16))
        ao = T("ao", lambda: torch.empty(Bp, d, device=dev, dtype=F16))
        gf = T("gf", lambda: torch.empty(Bp, dims.ffn, device=dev, dtype=F16))
        st = (T("st0", lambda: torch.empty(Bp, device=dev)), T("st1", lambda: torch.empty(Bp, device=dev)))
        logits = T("logits", lambda: torch.empty(Bp, Vp, device=dev, dtype=F16))
        if "sb" not in ws:
            ws["sb"] = _sequence_bias_tables(sequence_bias, V, dev)   # HF SequenceBiasLogitsProcessor (model.generate(sequence_bias=...))
        sb = ws["sb"]
        # processors + per-row top-k in one pass, no fp32 score matrix (a sequence bias takes the two-kernel form)
        scores = None if fused_select else T("scores", lambda: torch.empty(Bp, V, device=dev, dtype=F32))

        # (Row ranges of the batch as independent chains on separate streams -- VERDICT r4 #4 -- were built in round 5, gave identical ids and
        # were SLOWER: greedy 105.9 k tokens/s with one chain, 85.6 k with two, 74.6 k with four; a range's chain costs as many 4-9 us launches
        # whatever its row count.  Removed in round 6: tools/probe/attic/README.md, profiles/r5_probe_decode_split.log.)
        # row views made ONCE per generation: the loop below runs eagerly for short generations, and a dozen tensor slices per step
        # (~1.5 us each on the host) were enough to make a 0.84-ms step host-bound
        whole = dict(s0=0, s1=B, r0=0, n=Bp, h={id(t_): t_ for t_ in h}, x=x16, qkv=qkv, qc=qc, ao=ao, gf=gf, st=st,
                     anc={id(t_): t_ for t_ in anc}, kx=kvx, vt=vtx)

        def layers(pt: dict, t: int, a, c1):
            """the decoder layers for the sequences of one row range; results in h[0] / h[1] by layer-count parity"""
            s0, s1, r0, n = pt["s0"], pt["s1"], pt["r0"], pt["n"]
            hh = [pt["h"][id(h[0])], pt["h"][id(h[1])]]
            x_, qkv_, qc_, ao_, gf_, st_ = pt["x"], pt["qkv"], pt["qc"], pt["ao"], pt["gf"], pt["st"]
            a_ = pt["anc"][id(a)]
            for li, Lw in enumerate(eng.dec):
                ops.layernorm_fwd(hh[0], *Lw["ln1"], x_, *st_, n, d)
                eng._lin(x_, n, Lw["qkv"], C16=qkv_)
                # the kernel reads position t from this step's k | v rows and appends them to the cache itself
                ops.attn_decode(Q=qkv_, K=kvc[li], V=(kvc[li], d), O=ao_, groups=n, nq=1, H=H, Lk=t + 1, Lk_max=max_len,
                                ldq=3 * d, ldk=2 * d, ldv=2 * d, ldo=d, anc=a_, anc_ld=max_len, kv_pos_stride=Bp,
                                kv_len_dev=c1, Knew=(qkv_, d), Vnew=(qkv_, 2 * d), ldnew=3 * d, slot0=r0)
                eng._lin(ao_, n, Lw["out"], R32=hh[0], H32=hh[1])
                ops.layernorm_fwd(hh[1], *Lw["ln2"], x_, *st_, n, d)
                eng._lin(x_, n, Lw["cq"], C16=qc_)
                kx = pt["kx"][li]
                if fewq:
                    ops.attn_fewq(Q=qc_, K=kx, Vt=pt["vt"][li], O=ao_, groups=s1 - s0, nq=nb, H=H, Lk=S, ldq=d, ldk=2 * d,
                                  ldvt=Sp, ldo=d)
                elif nb > 1:
                    # beams of a sequence = the "queries" of one flash-attention problem over the sequence's encoder
                    # K/V: the MFMA kernel reads the 384 KB per (sequence, head) once and is HBM-bound (~60 us / layer
                    # at B = 128), where the per-key VALU dot products of ns_attn_decode took 108 us at 5 beams
                    ops.attn_fwd(Q=qc_, K=kx, V=(kx, d), O=ao_, B=s1 - s0, H=H, Lq=nb, Lk=S, ldq=d, ldk=2 * d,
                                 ldv=2 * d, ldo=d, causal=False)
                else:
                    ops.attn_decode(Q=qc_, K=kx, V=(kx, d), O=ao_, groups=s1 - s0, nq=nb, H=H, Lk=S, Lk_max=S, ldq=d,
                                    ldk=2 * d, ldv=2 * d, ldo=d, kv_group_stride=S)
                eng._lin(ao_, n, Lw["cout"], R32=hh[1], H32=hh[0])
                ops.layernorm_fwd(hh[0], *Lw["ln3"], x_, *st_, n, d)
                eng._lin(x_, n, Lw["fc1"], G16=gf_, gelu=True)     # (no pre-activation copy: nothing reads it without a backward)
                eng._lin(gf_, n, Lw["fc2"], R32=hh[0], H32=hh[1])
                hh.reverse()

        def step(tok: torch.Tensor, t: int, parent, ctr=None):
            """Feed token `tok` (Bp,) at position t; leaves last-position logits in `logits`.
            With `ctr` (device int32 [t, t+1]) the position is read on the device."""
            c0 = ctr
            c1 = (ctr, 1) if ctr is not None else None
            if nb > 1:
                ops.anc_update(anc[0], anc[1], parent, Bp, max_len, t, cur_dev=c0)
                anc.reverse()
            a = anc[0]
            ops.embed_pos(tok, eng.E32, eng.dec_pos, h[0], Bp, 1, d, pos0=t, pos0_dev=c0)
            layers(whole, t, a, c1)
            if len(eng.dec) % 2:
                h.reverse()
            ops.layernorm_fwd(h[0], *eng.dec_ln, x16, *st, Bp, d)
            ops.gemm(A=x16, am=rowmap(d), K=d, B=eng.E16, ldb=d, M=Bp, N=Vp, C16=logits, c16m=rowmap(Vp))

        for name, lst in (("suppress_tokens", suppress_tokens), ("begin_suppress_tokens", begin_suppress_tokens)):
            bad = [int(t) for t in lst if not 0 <= int(t) < V]
            if bad:     # these ids index the score row on the device
                raise ValueError(f"The model vocabulary size is {V}, but `{name}` holds {bad}")
        sup = T("sup", lambda: torch.tensor(list(suppress_tokens), device=dev, dtype=torch.int32)) if len(suppress_tokens) else None
        bsup = T("bsup", lambda: torch.tensor(list(begin_suppress_tokens), device=dev, dtype=torch.int32)) if len(begin_suppress_tokens) else None
        forced_tab, n_forced = None, 0
        if forced_decoder_ids:
            fmap = {int(i): t for i, t in forced_decoder_ids if t is not None}
            bad = [t for t in fmap.values() if not 0 <= int(t) < V]
            if bad:
                raise ValueError(f"The model vocabulary size is {V}, but `forced_decoder_ids` holds {bad}")
            if fmap:
                n_forced = max(fmap) + 1
                tab = [-1] * n_forced
                for i, t in fmap.items():
                    if i >= 0:
                        tab[i] = int(t)
                forced_tab = T("forced", lambda: torch.tensor(tab, device=dev, dtype=torch.int32))
        if forced_tab is not None:
            sb_forced = dict(forced=forced_tab, n_forced=n_forced)
        else:
            sb_forced = {}
        proc = dict(logits16=logits, scores32=scores, rows=Bp, V=V, ldv=Vp, ids_ld=max_len,
                    begin_index=P if begin_index is None else int(begin_index), **sb_forced,
                    repetition_penalty=float(repetition_penalty), no_repeat_ngram=int(no_repeat_ngram_size),
                    suppress=sup, n_suppress=len(suppress_tokens), begin_suppress=bsup,
                    n_begin_suppress=len(begin_suppress_tokens), **sb)

        seqs = [T(f"seqs{i}", lambda: torch.full((Bp, max_len), pad, device=dev, dtype=torch.int64), lambda t: t.fill_(pad)) for i in range(2)]
        seqs[0][:, :P] = prompt.repeat_interleave(nb, 0)
        for t in range(P):
            step(seqs[0][:, t].contiguous(), t, None)
        flags = T("flags", lambda: torch.zeros(2, device=dev, dtype=torch.int32), lambda t: t.zero_())
        next_tok = T("next_tok", lambda: torch.empty(Bp, device=dev, dtype=torch.int64))
        ctr_dev = T("ctr", lambda: torch.zeros(2, device=dev, dtype=torch.int32))
        cur = P

        graph_ok = self.use_graph and dev.type == "cuda" and trace is None

        def run_loop(select, ping_pong):
            """select(cur, ctr) picks token `cur` from `logits` (and reverses the lists in `ping_pong`); the step then
            feeds it at position `cur`.  First iteration eager (lazy kernel attributes / workspaces), then graphs."""
            nonlocal cur, graph_ok
            n_sel = 0
            ctr = None
            graphs = None
            is_list = False
            self.last_loop_mode = "eager"
            # host-bound watch (launch lists only): time spent inside the replays of a check_every-step chunk against the chunk's
            # wall time (the flag poll at its end synchronizes anyway)
            enq, chunk_t0, chunk_steps, slow_chunks = 0.0, None, 0, 0

            def feed():
                step(next_tok, cur, parent, ctr)
                ops.add_i32(ctr, 1, n=2)

            def build(as_list):
                """the two (select, feed) pairs -- one per ping-pong parity -- as launch lists or as hipGraphs.  Nothing is launched;
                the host-side lists end in the orientation they started in (two reversals)."""
                nonlocal graph_ok
                pairs = []
                torch.cuda.synchronize()
                # select() / step() reverse host-side ping-pong lists as they go: a capture that dies half way must leave them as they were
                snap = [(l_, list(l_)) for l_ in (anc, h, *ping_pong)]
                prev_stream = torch.cuda.current_stream() if dev.type == "cuda" else None
                gs = gt = None
                try:
                    for _ in range(2):
                        if as_list:
                            gs, gt = ops.LaunchList(), ops.LaunchList()
                            gs.keep = gt.keep = ws       # the session's tensors are what the recorded raw addresses point into
                            with ops.recording(gs):
                                select(cur, ctr)
                            with ops.recording(gt):
                                feed()
                        else:
                            gs, gt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                            # thread-local capture mode + the package's capture lock: other host threads (the data feed's
                            # loader thread synchronizes, allocates and copies on its own stream) must not invalidate this
                            with GPU_CAPTURE_LOCK:
                                with torch.cuda.graph(gs, capture_error_mode="thread_local"):
                                    select(cur, ctr)
                                with torch.cuda.graph(gt, capture_error_mode="thread_local"):
                                    feed()
                        pairs.append((gs, gt))
                except RuntimeError as e:
                    if as_list:
                        raise
                    # A host thread outside GPU_CAPTURE_LOCK can invalidate a capture on HIP (the rule of engine.train_step):
                    # nothing has executed.  torch.cuda.graph.__exit__ leaves its capture stream current when capture_end raises -- put
                    # the caller's back --, the half-built graph objects are never destroyed (their destructor has aborted the process
                    # on this stack), the session forgets its graphs, and ONE failure ends graph use for this Generator: the loop goes
                    # on with launch lists or eager launches.
                    torch.cuda.synchronize()
                    torch.cuda.set_stream(prev_stream)
                    for l_, saved in snap:
                        l_[:] = saved
                    self._graph_graveyard.append((pairs, gs, gt))
                    self.capture_failures += 1
                    self.use_graph = graph_ok = False
                    ws["graphs"] = None
                    import warnings
                    warnings.warn(f"generate: hipGraph capture failed ({str(e).splitlines()[0][:120]}); launch lists / eager launches from "
                                  "here on (graphs disabled for this Generator)")
                    return None
                return pairs

            def replay(g):
                nonlocal enq
                if is_list:
                    t0 = time.perf_counter()
                    g.replay()
                    enq += time.perf_counter() - t0
                else:
                    g.replay()

            while cur < max_len:
                if graphs is None:
                    select(cur, None)
                else:
                    replay(graphs[(n_sel - 1) & 1][0])
                n_sel += 1
                cur += 1
                if cur >= max_len:
                    break
                if (cur - P) % check_every == 0:
                    f = flags.tolist()
                    if f[0] == 0 or (nb > 1 and f[1] == 0):
                        break
                    if is_list:
                        # Launch lists save the descriptor building, not the runtime's own launch path: on a slow host the replay of
                        # ~72 launches still takes longer than the 0.86 ms the GPU needs for them (driver boxes of round 5: 1.00 ms per
                        # greedy step, 99 k tokens/s against 105-106 k).  Two chunks in a row in which the host spent > 95 % of the wall
                        # time inside the replays = host-bound: capture the hipGraphs after all (~3 ms, paid back within ~20 steps)
                        now = time.perf_counter()
                        if chunk_t0 is not None and chunk_steps == check_every:
                            slow_chunks = slow_chunks + 1 if enq > self.adaptive_frac * (now - chunk_t0) else 0
                            if slow_chunks >= 2 and graph_ok and self.adaptive and max_len - cur >= self.adaptive_min_steps:
                                g_new = build(False)
                                if g_new is not None:       # (a failed capture: the lists go on)
                                    graphs, is_list = g_new, False
                                    self.last_loop_mode = f"lists->graphs@{cur - P}"
                                    if key is not None:
                                        ws["graphs"] = graphs
                        enq, chunk_t0, chunk_steps = 0.0, time.perf_counter(), 0
                if graphs is None:
                    step(next_tok, cur - 1, parent)
                    # two chains per step double the launches the host has to enqueue (~150 per step against ~0.6 ms of GPU time):
                    # with split chains the replayed graph is what keeps the loop GPU-bound, so even short generations capture
                    gms = self.graph_min_steps
                    as_graph = graph_ok and max_len - cur >= gms
                    # Shorter generations replay LAUNCH LISTS instead (ops.LaunchList: the same two (select, feed) pairs recorded
                    # without being launched, exactly as a stream capture would; no hipGraph to instantiate): building ~80
                    # descriptors through ctypes costs the host 0.8-1.0 ms per step against 0.84 ms of GPU time -- the driver's
                    # 64-token eval leg ran host-bound on a slow host (100 k tokens/s against 108 k) -- and ~0.1 ms replayed.
                    as_list = not as_graph and self.use_lists and trace is None and max_len - cur >= 4
                    # A session that is called again (an evaluation loop: the same signature for every batch) replays what it holds --
                    # the hipGraphs, captured by the SECOND call of the signature whatever the generation's length, or the lists
                    if ws["graphs"] is None and graph_ok and key is not None and not first_call and max_len - cur >= 4:
                        as_graph = True
                    cached_graphs = ws["graphs"] is not None and graph_ok
                    if as_graph or as_list or cached_graphs:
                        # counters as of the NEXT iteration: it selects token `cur` and feeds it at position `cur`
                        ctr = ctr_dev
                        ctr.copy_(torch.tensor([cur, cur + 1], dtype=torch.int32))
                        if cached_graphs:
                            graphs, is_list = ws["graphs"], False
                            self.last_loop_mode = "graphs (session)"
                        elif as_graph and (g_new := build(False)) is not None:
                            graphs, is_list = g_new, False
                            self.last_loop_mode = "graphs"
                            if key is not None:
                                ws["graphs"] = graphs
                        elif not (as_list or (as_graph and self.use_lists and trace is None and max_len - cur >= 4)):
                            pass        # (the capture failed and lists are not an option here: eager launches)
                        elif ws["lists"] is not None:
                            graphs, is_list = ws["lists"], True
                            self.last_loop_mode = "lists (session)"
                        else:
                            graphs, is_list = build(True), True
                            self.last_loop_mode = "lists"
                            if key is not None:
                                ws["lists"] = graphs
                else:
                    replay(graphs[(n_sel - 2) & 1][1])
                    chunk_steps += 1
            if graphs is not None and (n_sel - 1) & 1:
                for pair in ping_pong:      # replays do not touch the host-side lists: re-apply the odd reversal
                    pair.reverse()

        if nb == 1:
            done = T("done", lambda: torch.zeros(Bp, device=dev, dtype=torch.uint8), lambda t: t.zero_())
            parent = None

            cand_v = T("cand_v", lambda: torch.empty(Bp, device=dev, dtype=F32))
            cand_i = T("cand_i", lambda: torch.empty(Bp, device=dev, dtype=torch.int32))

            if trace is not None and not fused_select:
                raise ValueError("trace needs the fused selection kernel (vocabulary within SELECT_MAX_LDV, no NS_NO_FUSED_SELECT)")
            tr_v = torch.empty(Bp, 2, device=dev, dtype=F32) if trace is not None else None
            tr_i = torch.empty(Bp, 2, device=dev, dtype=torch.int32) if trace is not None else None

            def select(c, ctr):
                ops.zero_(flags)
                if trace is not None:       # this step's two best processed scores per row, before the ids move on
                    ops.logits_select(ids=seqs[0], cur_len=c, log_softmax=False, cur_len_dev=ctr, k=2, group_rows=1,
                                      cand_vals=tr_v, cand_idx=tr_i, **proc)
                    trace.append((tr_v.clone(), tr_i.clone()))
                if fused_select:
                    ops.logits_select(ids=seqs[0], cur_len=c, log_softmax=False, cur_len_dev=ctr, k=1, group_rows=1,
                                      cand_vals=cand_v, cand_idx=cand_i, **proc)
                    ops.greedy_update(None, Bp, V, seqs[0], max_len, c, eos, pad, done, flags, next_tok, cur_dev=ctr,
                                      best_idx=cand_i)
                else:
                    ops.logits_process(ids=seqs[0], cur_len=c, log_softmax=False, cur_len_dev=ctr, **proc)
                    ops.greedy_update(scores, Bp, V, seqs[0], max_len, c, eos, pad, done, flags, next_tok, cur_dev=ctr)

            run_loop(select, [])
            out = seqs[0]
        else:
            run_scores = [T(f"run_scores{i}", lambda: torch.zeros(B, nb, device=dev, dtype=F32), lambda t: t.zero_()) for i in range(2)]
            run_scores[0][:, 1:] = -1e9
            fin_seqs = [T(f"fin_seqs{i}", lambda: seqs[0].clone(), lambda t: t.copy_(seqs[0])) for i in range(2)]
            fin_scores = [T(f"fin_scores{i}", lambda: torch.full((B, nb), -1e9, device=dev, dtype=F32), lambda t: t.fill_(-1e9)) for i in range(2)]
            fin_done = [T(f"fin_done{i}", lambda: torch.zeros(B, nb, device=dev, dtype=torch.uint8), lambda t: t.zero_()) for i in range(2)]
            open_row = T("open_row", lambda: torch.ones(B, device=dev, dtype=torch.uint8), lambda t: t.fill_(1))
            top_v = T("top_v", lambda: torch.empty(B, 2 * nb, device=dev, dtype=F32))
            top_i = T("top_i", lambda: torch.empty(B, 2 * nb, device=dev, dtype=torch.int32))
            parent = T("parent", lambda: torch.empty(Bp, device=dev, dtype=torch.int32))
            pairs = [seqs, run_scores, fin_seqs, fin_scores, fin_done]

            cand_v = T("cand_v", lambda: torch.empty(Bp, 2 * nb, device=dev, dtype=F32))
            cand_i = T("cand_i", lambda: torch.empty(Bp, 2 * nb, device=dev, dtype=torch.int32))

            def select(c, ctr):
                if fused_select:
                    ops.logits_select(ids=seqs[0], cur_len=c, log_softmax=True, beam_scores=run_scores[0], cur_len_dev=ctr,
                                      k=2 * nb, group_rows=nb, cand_vals=cand_v, cand_idx=cand_i, **proc)
                    ops.topk_merge(cand_v, cand_i, B, nb * 2 * nb, 2 * nb, top_v, top_i)
                else:
                    ops.logits_process(ids=seqs[0], cur_len=c, log_softmax=True, beam_scores=run_scores[0], cur_len_dev=ctr, **proc)
                    ops.topk_groups(scores, B, nb * V, 2 * nb, top_v, top_i)
                ops.zero_(flags)
                ops.beam_update(top_vals=top_v, top_idx=top_i, run_seqs_in=seqs[0], run_seqs_out=seqs[1],
                                run_scores_out=run_scores[1], fin_seqs_in=fin_seqs[0], fin_seqs_out=fin_seqs[1],
                                fin_scores_in=fin_scores[0], fin_scores_out=fin_scores[1], fin_done_in=fin_done[0],
                                fin_done_out=fin_done[1], open=open_row, parent_out=parent, next_tok_out=next_tok,
                                any_open=flags, any_continuation=(flags, 1), cur_len_dev=ctr, batch=B, num_beams=nb,
                                V=V, max_len=max_len, cur_len=c, prompt_len=P, eos_id=eos,
                                length_penalty=float(length_penalty))
                for pair in pairs:
                    pair.reverse()

            run_loop(select, pairs)
            out = fin_seqs[0].view(B, nb, max_len)[:, 0]
            self.last_scores = fin_scores[0][:, 0].clone()
        # crop like HF: up to the longest hypothesis (a hypothesis ends at its first EOS after the prompt)
        o = out.cpu()
        gen = o[:, P:]
        is_eos = gen == eos
        first = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, torch.full((o.shape[0],), gen.shape[1]))
        width = P + int(first.max().item()) if gen.shape[1] > 0 else P
        width = min(width, cur)
        ws["calls"] += 1
        res = out[:, :width].clone()       # (the session's buffers are written again by the next call)
        if key is not None and dev.type == "cuda":
            # sessions are a cache: together they may hold at most max_session_frac of the device memory (oldest out first; a single
            # session above the cap is not kept at all)
            cap = self.max_session_frac * torch.cuda.get_device_properties(dev).total_memory
            size = lambda w: sum(t.numel() * t.element_size() for t in w["t"].values())       # noqa: E731
            while self._sessions and sum(size(w) for w in self._sessions.values()) > cap:
                self._sessions.popitem(last=False)
        return res
