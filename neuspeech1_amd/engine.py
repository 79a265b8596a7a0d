"""The MI355X Whisper-MEG engine: explicit forward / backward / optimizer step of
NeuSpeech's training hot path as a fixed sequence of HIP kernel launches.

There is no autograd graph and no tracing compiler: the model is one known
architecture, so the backward pass is written out by hand against saved
activations, which lets gradients land directly in ONE flat fp32 buffer (what
RCCL all-reduces) in backward-completion order.

Reference path being replaced (file:line in the reference tree / HF):
  conv stem      utils/model_utils.py:9-23, utils/load_model.py:410-417
  encoder layer  HF:modeling_whisper.py:360-413 (+ :241-357 attention)
  decoder layer  HF:modeling_whisper.py:416-506
  LM head + CE   utils/load_model.py:1025-1054
  LoRA           finetune.py:187-212 (peft LoraConfig r=32, alpha=64, dropout .05)
  train step     finetune.py:231-253 (HF Trainer: fp16 autocast + GradScaler,
                 AdamW, clip 1.0, linear warmup/decay)

Numerics contract: fp32 residual stream, fp32 LayerNorm
/ softmax / CE statistics, fp16 GEMM operands with fp32 accumulation, fp16
rounding at every Linear/Conv output, exact-erf GELU evaluated on fp16 values.
"""
from __future__ import annotations

import os

import math
from dataclasses import dataclass, field

import torch

from . import ops
from .feed import PackedSignal
from .lib import GPU_CAPTURE_LOCK, AdamWCfg
from .ops import (NS_GEMM_ATOMIC32, NS_GEMM_COLSUM_A, NS_GEMM_DROP_A, NS_GEMM_GELU, NS_GEMM_GELU_SAVE_GRAD, NS_GEMM_MUL_P16, NS_GEMM_TN,
                  rowmap)

# Every GELU site saves gelu'(x) (fp16) where the reference keeps x for autograd: the backward seams then multiply
# (NS_GEMM_MUL_P16 / ns_dgelu_mul(pre_is_grad)) instead of re-evaluating erf / exp for every element.
GELU_FWD = NS_GEMM_GELU | NS_GEMM_GELU_SAVE_GRAD
_TN_TARGET = int(os.environ.get("NS_TN_TARGET", 768))     # workgroups a split weight-gradient launch aims at (A/B runs)
from .weights import LORA_SUFFIXES, WhisperDims

F16, F32 = torch.float16, torch.float32


@dataclass
class LoraSpec:
    """LoRA (finetune.py:210-211) or, with adalora=True, AdaLoRA at its initial rank (finetune.py:206-208: the rank
    allocator is never invoked by the reference's Trainer, so r stays init_r)."""
    r: int = 32
    alpha: float = 64.0
    dropout: float = 0.05
    adalora: bool = False
    orth_reg_weight: float = 0.5
    layers: int | None = None     # adapt only encoder layers 0 .. layers-1 (finetune.py --fine_tune_layers, :188-190)
    decoder: bool = False         # --ft_full (finetune.py:191-192): adapters on every decoder projection too

    @property
    def scale(self) -> float:
        return self.alpha / (self.r + 1e-5) if self.adalora else self.alpha / self.r

    @property
    def r_pad(self) -> int:
        return (self.r + 15) // 16 * 16


@dataclass
class TrainCfg:
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0
    max_grad_norm: float = 1.0
    warmup_steps: int = 0
    total_steps: int = 0
    fp16_scaler: bool = True          # GradScaler semantics (finetune.py:242)
    init_scale: float = 65536.0
    growth_factor: float = 2.0
    backoff_factor: float = 0.5
    growth_interval: int = 2000


# decoder adapter sites of one layer (--ft_full): (site, engine linear, projection names of the stacked groups,
# input width is d except fc2, output width per group, rows = encoder rows for the cross K/V)
def _dec_sites(d, f):
    return (("self_attn.qkv", "qkv", ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), d, d, False),
            ("self_attn.out_proj", "out", ("self_attn.out_proj",), d, d, False),
            ("encoder_attn.q_proj", "cq", ("encoder_attn.q_proj",), d, d, False),
            ("encoder_attn.kv", "ckv", ("encoder_attn.k_proj", "encoder_attn.v_proj"), d, d, True),
            ("encoder_attn.out_proj", "cout", ("encoder_attn.out_proj",), d, d, False),
            ("fc1", "fc1", ("fc1",), d, f, False),
            ("fc2", "fc2", ("fc2",), f, d, False))


class _Lin:
    """fp16 operand pack of a frozen Linear: W (N,K) for forward, W^T (K,N) for dgrad."""
    __slots__ = ("w", "wt", "bias", "N", "K")

    def __init__(self, w32: torch.Tensor, b32, scale_rows=None):
        w = w32.float()
        b = None if b32 is None else b32.float().clone()
        if scale_rows is not None:
            n, s = scale_rows
            w = w.clone()
            w[:n] *= s
            if b is not None:
                b[:n] *= s
        self.N, self.K = w.shape
        self.w = w.to(F16).contiguous()
        self.wt = w.t().to(F16).contiguous()
        self.bias = b


class _ResidentSignal:
    """a batch that already sits packed in an engine-owned `xin` buffer (train_step's graph path packs plain fp32 batches
    eagerly in front of the replay); quacks like feed.PackedSignal for encode()"""

    def __init__(self, xin, shape):
        self.xin, self.shape = xin, shape

    def acquire(self):
        return self


class MegWhisperEngine:
    def __init__(self, dims: WhisperDims, sd: dict, lora: LoraSpec | None = None, lora_sd: dict | None = None,
                 train_cfg: TrainCfg | None = None, device="cuda:0", train_convs: bool = True):
        self.dims, self.dev = dims, torch.device(device)
        self.lora = lora
        self.tc = train_cfg or TrainCfg()
        self.train_convs = train_convs
        d = dims.d
        assert d % 256 == 0 and d // dims.heads == 64, "kernels need d % 256 == 0 and head_dim 64"
        assert dims.src_pos % 4 == 0 and dims.src_pos >= 64
        g = lambda k: torch.as_tensor(sd[k]).to(self.dev, F32)  # noqa: E731
        self._sd_get = g
        s = 64 ** -0.5
        self.r = lora.r_pad if lora else 0          # MFMA K granularity: ranks are zero-padded to a multiple of 16
        self.r_real = lora.r if lora else 0
        self.adalora = bool(lora and lora.adalora)
        if self.adalora and lora.r > 32:
            raise ValueError(f"AdaLoRA init_r={lora.r}: ns_orth_reg handles ranks up to 32 (NS_ORTH_MAX_R)")
        # front-end variant (utils/model_utils.py:9-23): 'base' = conv1.0 (k3,s1) + GELU + conv1.2 (k3,s2); 'replace' =
        # ONE stride-2 conv `encoder.conv1` straight from the MEG channels
        self.frontend = "replace" if ("model.encoder.conv1.weight" in sd and "model.encoder.conv1.0.weight" not in sd) else "base"
        self.n_lora = 0 if not lora else (dims.enc_layers if lora.layers is None else int(lora.layers))
        self.dec_lora = bool(lora and lora.decoder)
        if self.dec_lora and self.n_lora != dims.enc_layers:
            raise ValueError("LoraSpec.decoder (--ft_full) adapts the whole model: it excludes LoraSpec.layers")
        if lora and not 1 <= self.n_lora <= dims.enc_layers:
            raise ValueError(f"LoraSpec.layers={lora.layers} outside 1..{dims.enc_layers}")
        # ---------------- frozen operand packs
        e = "model.encoder."
        self.enc_pos = g(e + "embed_positions.weight").contiguous()
        self.enc = []
        for i in range(dims.enc_layers):
            p = f"{e}layers.{i}."
            zk = torch.zeros(d, device=self.dev)
            L = {}
            L["qkv"] = _Lin(torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"),
                                       g(p + "self_attn.v_proj.weight")], 0),
                            torch.cat([g(p + "self_attn.q_proj.bias"), zk, g(p + "self_attn.v_proj.bias")], 0), (d, s))
            L["out"] = _Lin(g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
            L["fc1"] = _Lin(g(p + "fc1.weight"), g(p + "fc1.bias"))
            L["fc2"] = _Lin(g(p + "fc2.weight"), g(p + "fc2.bias"))
            L["ln1"] = (g(p + "self_attn_layer_norm.weight"), g(p + "self_attn_layer_norm.bias"))
            L["ln2"] = (g(p + "final_layer_norm.weight"), g(p + "final_layer_norm.bias"))
            self.enc.append(L)
        self.enc_ln = (g(e + "layer_norm.weight"), g(e + "layer_norm.bias"))
        dd = "model.decoder."
        self.E32 = g(dd + "embed_tokens.weight").contiguous()
        self.dec_pos = g(dd + "embed_positions.weight").contiguous()
        Ep = torch.zeros(dims.vocab_pad, d, device=self.dev, dtype=F16)
        Ep[:dims.vocab] = self.E32.to(F16)
        self.E16 = Ep
        self.E16T = Ep.t().contiguous()
        self.dec = []
        for i in range(dims.dec_layers):
            p = f"{dd}layers.{i}."
            zk = torch.zeros(d, device=self.dev)
            L = {}
            L["qkv"] = _Lin(torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"),
                                       g(p + "self_attn.v_proj.weight")], 0),
                            torch.cat([g(p + "self_attn.q_proj.bias"), zk, g(p + "self_attn.v_proj.bias")], 0), (d, s))
            L["out"] = _Lin(g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
            L["cq"] = _Lin(g(p + "encoder_attn.q_proj.weight"), g(p + "encoder_attn.q_proj.bias"), (d, s))
            L["ckv"] = _Lin(torch.cat([g(p + "encoder_attn.k_proj.weight"), g(p + "encoder_attn.v_proj.weight")], 0),
                            torch.cat([zk, g(p + "encoder_attn.v_proj.bias")], 0))
            L["cout"] = _Lin(g(p + "encoder_attn.out_proj.weight"), g(p + "encoder_attn.out_proj.bias"))
            L["fc1"] = _Lin(g(p + "fc1.weight"), g(p + "fc1.bias"))
            L["fc2"] = _Lin(g(p + "fc2.weight"), g(p + "fc2.bias"))
            L["ln1"] = (g(p + "self_attn_layer_norm.weight"), g(p + "self_attn_layer_norm.bias"))
            L["ln2"] = (g(p + "encoder_attn_layer_norm.weight"), g(p + "encoder_attn_layer_norm.bias"))
            L["ln3"] = (g(p + "final_layer_norm.weight"), g(p + "final_layer_norm.bias"))
            self.dec.append(L)
        self.dec_ln = (g(dd + "layer_norm.weight"), g(dd + "layer_norm.bias"))
        # The cross-attention K | V projections of ALL decoder layers read the same encoder states (HF:modeling_whisper.py:279-282 with
        # key_value_states = encoder_hidden_states in every layer): in training they run as ONE GEMM with the layers' weights stacked
        # (N = layers x 2d) and their input gradients as ONE GEMM over K = layers x 2d -- the encoder rows are streamed once instead of
        # once per layer, and the fp32 encoder-state gradient is written once instead of being read, added to and rewritten per layer.
        # NS_CKV_BATCH=0 keeps the per-layer launches (A/B runs); decoder adapters (--ft_full) always do.
        self.ckv_batch = os.environ.get("NS_CKV_BATCH", "1") != "0" and not self.dec_lora and dims.dec_layers > 1
        if self.ckv_batch:
            self.ckv_all = _Lin(torch.cat([L["ckv"].w.float() for L in self.dec], 0), torch.cat([L["ckv"].bias for L in self.dec], 0))
        self._build_trainables(sd, lora_sd)
        self._bufs = {}
        # hipGraph replay of train_step (see there); NS_TRAIN_GRAPH=0 keeps every step eager
        self.use_graph = os.environ.get("NS_TRAIN_GRAPH", "1") != "0"
        self._graphs, self._graph_warm, self.graph_cache = {}, set(), 48
        self._graph_captures = self._graph_failed_total = self._graph_replays = self._graph_evictions = 0
        self._graph_graveyard = []
        self.label_pad = int(os.environ.get("NS_LABEL_PAD", 16))
        self.no_fused_lora_bwd = False      # tests / A-B runs: keep the two-GEMM backward of the adapter up-projections
        self.no_side_u2 = os.environ.get("NS_NO_SIDE_U2") == "1"   # tests / A-B runs: fc2's adapter bottleneck by its own pass over the GELU output
        # residual Linear + the LayerNorm that reads its result in one launch (ns_gemm_ln: d = 512, bitwise the two launches' results).
        # NS_ROWLN: 0 = ns_gemm + ns_layernorm_fwd everywhere, 1 = every residual Linear with d = 512, 2 (default) = only the K = 512
        # out-projection.  Measured (round 5, same box, M = 96 000): K = 512 163.5 us against 166.6 us for the two launches; K = 2048 (fc2)
        # 314.6 against 277.7 us -- the pair's second pass costs less than this kernel's slower main loop there; whole step 31.63 (mode 2) /
        # 31.75-31.85 (mode 1) / 31.63-31.67 ms (mode 0): mode 2 takes six launches out of the step at the same time.
        self.use_rowln = int(os.environ.get("NS_ROWLN", "2"))
        self._init_opt_state()

    # ------------------------------------------------------------------ trainables
    def _conv_list(self):
        """(name, in channels, padded in channels) of the trainable convs, in backward-completion order"""
        d = self.dims.d
        if self.frontend == "base":
            return (("conv2", d, d), ("conv1.2", d, d), ("conv1.0", self.dims.ch, self.dims.ch_pad))
        return (("conv2", d, d), ("conv1", self.dims.ch, self.dims.ch_pad))

    def _build_trainables(self, sd, lora_sd):
        """One flat fp32 buffer, ordered by backward completion: LoRA of the top encoder layer first, conv stem last."""
        dims, d, f, r = self.dims, self.dims.d, self.dims.ffn, self.r
        Cp = dims.ch_pad
        segs = []  # (name, numel)
        self.lora_names = []
        if self.dec_lora:       # the decoder's backward runs first: its adapter gradients are final first
            for i in reversed(range(dims.dec_layers)):
                p = f"model.decoder.layers.{i}."
                for site, _, projs, kin, nout, _ in _dec_sites(d, f):
                    G = len(projs)
                    segs.append((p + site + ".lora_A", G * r * kin))
                    segs += [(p + pj + ".lora_B", nout * r) for pj in projs]
                    if self.adalora:
                        segs.append((p + site + ".lora_E", G * r))
        if self.lora:
            for i in reversed(range(self.n_lora)):
                p = f"model.encoder.layers.{i}."
                # A of q,k,v contiguous -> one stacked (3r, d) operand
                segs += [(p + "self_attn.qkv.lora_A", 3 * r * d)]
                for nm in ("q_proj", "k_proj", "v_proj"):
                    segs.append((p + f"self_attn.{nm}.lora_B", d * r))
                segs += [(p + "self_attn.out_proj.lora_A", r * d), (p + "self_attn.out_proj.lora_B", d * r),
                         (p + "fc1.lora_A", r * d), (p + "fc1.lora_B", f * r),
                         (p + "fc2.lora_A", r * f), (p + "fc2.lora_B", d * r)]
                if self.adalora:
                    segs += [(p + "self_attn.qkv.lora_E", 3 * r), (p + "self_attn.out_proj.lora_E", r),
                             (p + "fc1.lora_E", r), (p + "fc2.lora_E", r)]
        segs += [("model.encoder.conv2.wp", d * 3 * d), ("model.encoder.conv2.bias", d)]
        if self.frontend == "base":
            segs += [("model.encoder.conv1.2.wp", d * 3 * d), ("model.encoder.conv1.2.bias", d),
                     ("model.encoder.conv1.0.wp", d * 3 * Cp), ("model.encoder.conv1.0.bias", d)]
        else:
            segs += [("model.encoder.conv1.wp", d * 3 * Cp), ("model.encoder.conv1.bias", d)]
        self.seg_off = {}
        off = 0
        for nm, n in segs:
            self.seg_off[nm] = (off, n)
            off += (n + 63) // 64 * 64
        self.n_train = off
        self.P = torch.zeros(off, device=self.dev, dtype=F32)
        self.G = torch.zeros(off, device=self.dev, dtype=F32)
        self.lora_end = self.seg_off["model.encoder.conv2.wp"][0]
        g = self._sd_get
        # conv masters in packed GEMM layout (N, 3, Cp): wp[n, k, c] = w[n, c, k]
        for nm, cin, cp in self._conv_list():
            w = g(f"model.encoder.{nm}.weight")
            wp = self.pview(f"model.encoder.{nm}.wp").view(d, 3, cp)
            wp[:, :, :cin] = w.permute(0, 2, 1)
            self.pview(f"model.encoder.{nm}.bias").copy_(g(f"model.encoder.{nm}.bias"))
        if self.lora:
            rr = self.r_real

            def init_pair(a_view, b_view, e_view, key):
                """a_view (r_pad, in), b_view (out, r_pad), e_view (r_pad,) or None; only the first rr ranks are live."""
                if lora_sd is not None:
                    a_view[:rr].copy_(torch.as_tensor(lora_sd[key + ".lora_A.weight"]))
                    b_view[:, :rr].copy_(torch.as_tensor(lora_sd[key + ".lora_B.weight"]))
                    if e_view is not None:
                        e_view[:rr].copy_(torch.as_tensor(lora_sd[key + ".lora_E.weight"]).reshape(-1))
                elif self.adalora:      # peft AdaLoraLayer.reset_lora_parameters: A, B ~ N(0, 0.02), E = 0
                    a_view[:rr].normal_(0.0, 0.02)
                    b_view[:, :rr].normal_(0.0, 0.02)
                else:                   # peft LoRA: A kaiming_uniform(a=sqrt(5)), B = 0
                    torch.nn.init.kaiming_uniform_(a_view[:rr], a=math.sqrt(5))

            for i in range(self.n_lora):
                p = f"model.encoder.layers.{i}."
                A = self.pview(p + "self_attn.qkv.lora_A").view(3, r, d)
                E3 = self.pview(p + "self_attn.qkv.lora_E").view(3, r) if self.adalora else None
                for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                    init_pair(A[j], self.pview(p + f"self_attn.{nm}.lora_B").view(d, r), E3[j] if self.adalora else None,
                              p + f"self_attn.{nm}")
                for nm, (no, ki) in (("self_attn.out_proj", (d, d)), ("fc1", (f, d)), ("fc2", (d, f))):
                    init_pair(self.pview(p + nm + ".lora_A").view(r, ki), self.pview(p + nm + ".lora_B").view(no, r),
                              self.pview(p + nm + ".lora_E") if self.adalora else None, p + nm)
            if self.dec_lora:
                for i in range(dims.dec_layers):
                    p = f"model.decoder.layers.{i}."
                    for site, _, projs, kin, nout, _ in _dec_sites(d, f):
                        G = len(projs)
                        A = self.pview(p + site + ".lora_A").view(G, r, kin)
                        E = self.pview(p + site + ".lora_E").view(G, r) if self.adalora else None
                        for j, pj in enumerate(projs):
                            init_pair(A[j], self.pview(p + pj + ".lora_B").view(nout, r), E[j] if self.adalora else None, p + pj)
        self._build_operand_copies()

    def pview(self, name):
        o, n = self.seg_off[name]
        return self.P[o:o + n]

    def gview(self, name):
        o, n = self.seg_off[name]
        return self.G[o:o + n]

    def _build_operand_copies(self):
        """fp16 operand copies of the trainables + the device job table that refreshes them after each step."""
        dims, d, f, r = self.dims, self.dims.d, self.dims.ffn, self.r
        Cp = dims.ch_pad
        jobs = []
        z = lambda *s: torch.zeros(*s, device=self.dev, dtype=F16)  # noqa: E731
        P = self.P
        pp = lambda name: P.data_ptr() + 4 * self.seg_off[name][0]  # noqa: E731
        self.conv_ops = {}
        for nm, _, cp in self._conv_list():
            key = f"model.encoder.{nm}"
            o = {"w": z(d, 3 * cp), "cp": cp}
            jobs.append((pp(key + ".wp"), o["w"].data_ptr(), d, 3 * cp, 3 * cp, 3 * cp, 1.0, 0))
            if nm in ("conv2", "conv1.2"):  # stride-2 dgrad operands: even rows use tap 1, odd rows taps (2 | 0)
                o["we"] = z(cp, d)
                o["wo"] = z(cp, 2 * d)
                jobs.append((pp(key + ".wp") + 4 * cp, o["we"].data_ptr(), d, cp, 3 * cp, d, 1.0, 1))
                jobs.append((pp(key + ".wp") + 4 * 2 * cp, o["wo"].data_ptr(), d, cp, 3 * cp, 2 * d, 1.0, 1))
                jobs.append((pp(key + ".wp"), o["wo"].data_ptr() + 2 * d, d, cp, 3 * cp, 2 * d, 1.0, 1))
            self.conv_ops[nm] = o
        self.lora_ops = []
        if self.lora:
            sc = self.lora.scale
            qs = 64 ** -0.5
            for i in range(self.n_lora):
                p = f"model.encoder.layers.{i}."
                o = {"Aqkv": z(3 * r, d), "AqkvT": z(d, 3 * r), "sBqkv": z(3 * d, r),
                     "sBqT": z(r, d), "sBkT": z(r, d), "sBvT": z(r, d)}
                jobs.append((pp(p + "self_attn.qkv.lora_A"), o["Aqkv"].data_ptr(), 3 * r, d, d, d, 1.0, 0))
                jobs.append((pp(p + "self_attn.qkv.lora_A"), o["AqkvT"].data_ptr(), 3 * r, d, d, 3 * r, 1.0, 1))
                for j, (nm, s_) in enumerate((("q_proj", sc * qs), ("k_proj", sc), ("v_proj", sc))):
                    src = pp(p + f"self_attn.{nm}.lora_B")
                    ecol = pp(p + "self_attn.qkv.lora_E") + 4 * j * r if self.adalora else 0   # diag(E) folded into B
                    jobs.append((src, o["sBqkv"].data_ptr() + 2 * j * d * r, d, r, r, r, s_, 0, ecol))
                    jobs.append((src, o[("sBqT", "sBkT", "sBvT")[j]].data_ptr(), d, r, r, d, s_, 1, ecol))
                for nm, key, no, ki in (("self_attn.out_proj", "out", d, d), ("fc1", "fc1", f, d), ("fc2", "fc2", d, f)):
                    o[key + "_A"] = z(r, ki)
                    o[key + "_AT"] = z(ki, r)
                    o[key + "_sB"] = z(no, r)
                    o[key + "_sBT"] = z(r, no)
                    jobs.append((pp(p + nm + ".lora_A"), o[key + "_A"].data_ptr(), r, ki, ki, ki, 1.0, 0))
                    jobs.append((pp(p + nm + ".lora_A"), o[key + "_AT"].data_ptr(), r, ki, ki, r, 1.0, 1))
                    ecol = pp(p + nm + ".lora_E") if self.adalora else 0
                    jobs.append((pp(p + nm + ".lora_B"), o[key + "_sB"].data_ptr(), no, r, r, r, sc, 0, ecol))
                    jobs.append((pp(p + nm + ".lora_B"), o[key + "_sBT"].data_ptr(), no, r, r, no, sc, 1, ecol))
                self.lora_ops.append(o)
        self.dec_ads = []
        if self.dec_lora:
            sc, qs = self.lora.scale, 64 ** -0.5
            for i in range(dims.dec_layers):
                p = f"model.decoder.layers.{i}."
                ads = {}
                for site, lin, projs, kin, nout, enc_rows in _dec_sites(d, f):
                    G = len(projs)
                    ad = {"site": p + site, "projs": [p + pj for pj in projs], "G": G, "kin": kin, "nout": nout,
                          "enc_rows": enc_rows, "A": z(G * r, kin), "AT": z(kin, G * r), "sB": z(G * nout, r),
                          "sBT": [z(r, nout) for _ in range(G)],
                          "alpha": [sc * qs if pj.endswith("q_proj") else sc for pj in projs]}
                    jobs.append((pp(p + site + ".lora_A"), ad["A"].data_ptr(), G * r, kin, kin, kin, 1.0, 0))
                    jobs.append((pp(p + site + ".lora_A"), ad["AT"].data_ptr(), G * r, kin, kin, G * r, 1.0, 1))
                    for j, pj in enumerate(projs):
                        src = pp(p + pj + ".lora_B")
                        ecol = pp(p + site + ".lora_E") + 4 * j * r if self.adalora else 0
                        jobs.append((src, ad["sB"].data_ptr() + 2 * j * nout * r, nout, r, r, r, ad["alpha"][j], 0, ecol))
                        jobs.append((src, ad["sBT"][j].data_ptr(), nout, r, r, nout, ad["alpha"][j], 1, ecol))
                    ads[lin] = ad
                self.dec_ads.append(ads)
        self._job_table, self._njobs = ops.make_cast_jobs(jobs, self.dev)
        self._orth_table = None
        if self.adalora:
            # orthogonality regulariser over every lora_A (r x in) and lora_B (out x r), live ranks only
            oj, rr = [], self.r_real
            gp = lambda name: self.G.data_ptr() + 4 * self.seg_off[name][0]  # noqa: E731
            for i in range(self.n_lora):
                p = f"model.encoder.layers.{i}."
                for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                    oj.append((pp(p + "self_attn.qkv.lora_A") + 4 * j * r * d, gp(p + "self_attn.qkv.lora_A") + 4 * j * r * d, rr, d, d, 0))
                    oj.append((pp(p + f"self_attn.{nm}.lora_B"), gp(p + f"self_attn.{nm}.lora_B"), rr, d, r, 1))
                for nm, no, ki in (("self_attn.out_proj", d, d), ("fc1", f, d), ("fc2", d, f)):
                    oj.append((pp(p + nm + ".lora_A"), gp(p + nm + ".lora_A"), rr, ki, ki, 0))
                    oj.append((pp(p + nm + ".lora_B"), gp(p + nm + ".lora_B"), rr, no, r, 1))
            for ads in self.dec_ads:
                for ad in ads.values():
                    G, kin, nout = ad["G"], ad["kin"], ad["nout"]
                    for j, pj in enumerate(ad["projs"]):
                        oj.append((pp(ad["site"] + ".lora_A") + 4 * j * r * kin, gp(ad["site"] + ".lora_A") + 4 * j * r * kin, rr, kin, kin, 0))
                        oj.append((pp(pj + ".lora_B"), gp(pj + ".lora_B"), rr, nout, r, 1))
            self._orth_table, self._n_orth = ops.make_orth_jobs(oj, self.dev)
            # fused adapter backward: the folded operands' gradients of one layer land in ONE scratch buffer (q | k | v, out,
            # fc1, fc2 at fixed offsets) and a single ns_adalora_fold_jobs launch per layer turns them into dB / dE and
            # clears the scratch again (36 fold launches + 24 fills per step before)
            self._ada_off = {"qkv": 0, "out": 3 * d * r, "fc1": 4 * d * r, "fc2": (4 * d + f) * r}
            self._ada_scr = torch.zeros((5 * d + f) * r, device=self.dev, dtype=F32)
            self._fold_tables = []
            sc_, qs_ = self.lora.scale, 64 ** -0.5
            pa = lambda name, off=0: self.P.data_ptr() + 4 * (self.seg_off[name][0] + off)  # noqa: E731
            ga = lambda name, off=0: self.G.data_ptr() + 4 * (self.seg_off[name][0] + off)  # noqa: E731
            for i in range(self.n_lora):
                p = f"model.encoder.layers.{i}."
                fj = []
                for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                    k = p + f"self_attn.{nm}.lora_B"
                    fj.append((self._ada_scr.data_ptr() + 4 * (self._ada_off["qkv"] + j * d * r), pa(k), pa(p + "self_attn.qkv.lora_E", j * r),
                               ga(k), ga(p + "self_attn.qkv.lora_E", j * r), d, r, sc_ * qs_ if j == 0 else sc_))
                for nm, slot, no in (("self_attn.out_proj", "out", d), ("fc1", "fc1", f), ("fc2", "fc2", d)):
                    k = p + nm
                    fj.append((self._ada_scr.data_ptr() + 4 * self._ada_off[slot], pa(k + ".lora_B"), pa(k + ".lora_E"), ga(k + ".lora_B"),
                               ga(k + ".lora_E"), no, r, sc_))
                self._fold_tables.append(ops.make_fold_jobs(fj, self.dev))
            self.reg_dev = torch.zeros(1, device=self.dev)
            self.total_loss_dev = torch.zeros(1, device=self.dev)
            self._gbf = torch.zeros(max(d, f) * r, device=self.dev, dtype=F32)
            self._gbf3 = torch.zeros(max(3 * d, f) * r, device=self.dev, dtype=F32)
        self.refresh_operands()

    def refresh_operands(self):
        ops.cast_jobs(self._job_table, self._njobs)

    def _init_opt_state(self):
        n, dev = self.n_train, self.dev
        self.M1 = torch.zeros(n, device=dev)
        self.M2 = torch.zeros(n, device=dev)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.norm2_dev = torch.zeros(1, device=dev)
        self.found_inf_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.loss_scale_dev = torch.tensor([self.tc.init_scale if self.tc.fp16_scaler else 1.0], device=dev)
        self.growth_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.norm_ws = torch.empty(8192, device=dev, dtype=torch.uint8)
        self.loss_dev = torch.zeros(1, device=dev)
        self.nvalid_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        # LoRA-dropout seeds: `drop_seed` is the run's (per-rank) base and never changes on the host; what makes step t's
        # masks differ from step t-1's is a DEVICE counter that optimizer_step advances and every masked launch reads
        # (ns_gemm_desc.seed_dev) -- no launch argument varies from step to step, so a captured step replays unchanged
        self.drop_seed = 0x1234
        self.seed_ctr = torch.zeros(1, device=dev, dtype=torch.int32)

    # ------------------------------------------------------------------ buffers
    def _alloc(self, B: int, L: int, train: bool):
        """Activation / gradient buffers of one (batch, label length).  Everything sized by the ENCODER rows (B * 1500:
        ~17 GB at B = 64) is keyed by the batch size alone and shared by every label length; only the decoder-row buffers
        (B * L rows, ~0.6 GB) exist per length -- label lengths vary from batch to batch in the reference's recipes."""
        key = (B, L, train)
        if key in self._bufs:
            return self._bufs[key]
        b = dict(self._alloc_enc(B, train))
        if L > 0:
            b.update(self._alloc_dec(B, L, train))
        b["L"] = L
        self._bufs[key] = b
        return b

    def _alloc_enc(self, B: int, train: bool):
        key = (B, "enc", train)
        if key in self._bufs:
            return self._bufs[key]
        dims, d, f, r, H = self.dims, self.dims.d, self.dims.ffn, self.r, self.dims.heads
        T, S, Cp = dims.T, dims.src_pos, dims.ch_pad
        M = B * S
        dev = self.dev
        h16 = lambda *s: ops.zeros(*s, device=dev, dtype=F16)  # noqa: E731  (zeros: halo rows must be 0; cleared by ns_zero_spans)
        f32 = lambda *s: ops.zeros(*s, device=dev, dtype=F32)  # noqa: E731
        b = {"B": B}
        b["xin"] = h16(B, T + 2, Cp)
        if self.frontend == "base":
            b["pre0"] = h16(B * T, d)
            b["g0"] = h16(B, T + 2, d)
        b["pre1"] = h16(B * T // 2, d)
        b["g1"] = h16(B, T // 2 + 2, d)
        b["pre2"] = h16(M, d)
        ne = dims.enc_layers
        b["h"] = [f32(M, d) for _ in range((2 * ne + 1) if train else 2)]
        nl = ne if train else 1
        b["x1"] = [h16(M, d) for _ in range(nl)]
        b["x2"] = [h16(M, d) for _ in range(nl)]
        b["st1"] = [(f32(M), f32(M)) for _ in range(nl)]
        b["st2"] = [(f32(M), f32(M)) for _ in range(nl)]
        b["qkv"] = [h16(M, 3 * d) for _ in range(nl)]
        b["ao"] = [h16(M, d) for _ in range(nl)]
        b["lse"] = [f32(B, H, S) for _ in range(nl)]
        b["pre_f"] = [h16(M, f) for _ in range(nl)]
        b["gf"] = [h16(M, f) for _ in range(nl)]
        if r:
            b["uqkv"] = [h16(M, 3 * r) for _ in range(nl)]
            b["uo"] = [h16(M, r) for _ in range(nl)]
            b["u1"] = [h16(M, r) for _ in range(nl)]
            b["u2"] = [h16(M, r) for _ in range(nl)]
            # fc2's adapter bottleneck as a side product of fc1's GELU epilogue (ns_gemm_desc.side_*): one (M x 32) fp32 slab per
            # 256-column tile of the GELU output, summed by ns_gemm_side_reduce
            if r == 32 and ops.gemm_side_supported(M, f, d) and not self.no_side_u2:
                b["u2_slabs"] = torch.empty((f // 256) * M * 32, device=dev, dtype=F32)
        b["enc16"] = h16(M, d)
        b["enc_st"] = (f32(M), f32(M))
        ndl = dims.dec_layers if train else 1
        # cross-attention K | V of the encoder rows, per decoder layer: (tensor, column offset) + row stride b["kv_ld"]
        # (only while the stacked buffer stays below 2 GiB: the 256 x 256 GEMM addresses its operands with 32-bit byte offsets, and the input
        # gradient reads the stacked d(K|V) as its A operand -- whisper-large-v2's 32 layers x 2560 columns would fall back to slower kernels)
        if train and self.ckv_batch and M * ndl * 2 * d * 2 < 0x7FFF0000:
            b["kv_all"] = h16(M, ndl * 2 * d)
            b["kv_c"] = [(b["kv_all"], i * 2 * d) for i in range(ndl)]
            b["kv_ld"] = ndl * 2 * d
        else:
            b["kv_c"] = [(h16(M, 2 * d), 0) for _ in range(ndl)]
            b["kv_ld"] = 2 * d
        if self.dec_lora:
            b["ud_ckv"] = [h16(M, 2 * r) for _ in range(ndl)]
        if train:
            b["dh32"] = f32(M, d)
            b["dh16"] = h16(M, d)
            b["dx16"] = h16(M, d)
            b["dqkv"] = h16(M, 3 * d)
            b["dpre_f"] = h16(M, f)
            b["dao"] = h16(M, d)
            b["delta"] = f32(B, H, S)
            # fp32 scratch of the one-pass attention backward (dQ summed over a head's key sweeps; ns_attn_bwd1.hip)
            nws = ops.attn_bwd_workspace_bytes(B, H, S, S) if os.environ.get("NS_ATTN_TWO_PASS") != "1" else 0
            b["attn_ws"] = torch.empty(nws, device=dev, dtype=torch.uint8) if nws else None
            b["denc32"] = f32(M, d)
            if "kv_all" in b:
                b["dkv_all"] = h16(M, ndl * 2 * d)
                b["dkv_c"] = [(b["dkv_all"], i * 2 * d) for i in range(ndl)]
            else:
                b["dkv_c"] = [(h16(M, 2 * d), 0)] * ndl
            if r:
                b["du3"] = h16(M, 3 * r)
                b["du"] = h16(M, r)
            if self.dec_lora:
                b["ddu_kv"] = h16(M, 2 * r)
            b["dpre2"] = h16(B, S + 2, d)
            b["dpre1"] = h16(B, T // 2 + 2, d)
            b["dpre0"] = h16(B * T, d)
        self._bufs[key] = b
        return b

    def _alloc_dec(self, B: int, L: int, train: bool):
        dims, d, f, r, H = self.dims, self.dims.d, self.dims.ffn, self.r, self.dims.heads
        ML = B * L
        dev = self.dev
        h16 = lambda *s: ops.zeros(*s, device=dev, dtype=F16)  # noqa: E731
        f32 = lambda *s: ops.zeros(*s, device=dev, dtype=F32)  # noqa: E731
        nd = dims.dec_layers
        ndl = nd if train else 1
        b = {}
        b["hd"] = [f32(ML, d) for _ in range((3 * nd + 1) if train else 2)]
        for k in ("xs", "xc", "xm", "ao_s", "ao_c", "q_c"):
            b[k] = [h16(ML, d) for _ in range(ndl)]
        for k in ("st_s", "st_c", "st_m"):
            b[k] = [(f32(ML), f32(ML)) for _ in range(ndl)]
        b["qkv_s"] = [h16(ML, 3 * d) for _ in range(ndl)]
        b["lse_s"] = [f32(B, H, L) for _ in range(ndl)]
        b["lse_c"] = [f32(B, H, L) for _ in range(ndl)]
        b["pre_fd"] = [h16(ML, f) for _ in range(ndl)]
        b["gfd"] = [h16(ML, f) for _ in range(ndl)]
        if self.dec_lora:
            for lin, G in (("qkv", 3), ("out", 1), ("cq", 1), ("cout", 1), ("fc1", 1), ("fc2", 1)):
                b["ud_" + lin] = [h16(ML, G * r) for _ in range(ndl)]
        b["xd"] = h16(ML, d)
        b["st_d"] = (f32(ML), f32(ML))
        b["logits"] = h16(ML, dims.vocab_pad)
        b["row_loss"] = f32(ML)
        b["dec_ids"] = ops.zeros(B, L, device=dev, dtype=torch.int64)
        if train:
            if self.dec_lora:
                b["ddu"] = h16(ML, 3 * r)
            b["ddh32"] = f32(ML, d)
            b["ddx32"] = f32(ML, d)
            b["ddh16"] = h16(ML, d)
            b["ddx16"] = h16(ML, d)
            b["ddqkv"] = h16(ML, 3 * d)
            b["ddq_c"] = h16(ML, d)
            b["ddao"] = h16(ML, d)
            b["ddpre_f"] = h16(ML, f)
            b["ddelta"] = f32(B, H, L)
            # cross-attention backward in one pass (few queries): fp32 dQ slabs, one per group of key blocks (0 bytes: the two-pass kernels)
            nws = ops.attn_bwd_workspace_bytes(B, H, L, dims.src_pos) if os.environ.get("NS_ATTN_TWO_PASS") != "1" else 0
            b["attn_ws_c"] = torch.empty(nws, device=dev, dtype=torch.uint8) if nws else None
        return b

    # ------------------------------------------------------------------ helpers
    def _gemm(self, **kw):
        """ns_gemm with this engine's device-resident dropout counter attached"""
        if kw.get("drop_p", 0.0) > 0.0 or kw.get("side_drop_p", 0.0) > 0.0:
            kw["seed_dev"] = self.seed_ctr
        ops.gemm(**kw)

    def _lin(self, x16, M, lin: _Lin, *, C16=None, ldc=None, G16=None, R32=None, H32=None, gelu=False,
             A2=None, lda2=0, K2=0, B2=None, ngroup=0, **side):
        self._gemm(A=x16, am=rowmap(lin.K), K=lin.K, B=lin.w, ldb=lin.K, M=M, N=lin.N, bias=lin.bias, **side,
                 A2=A2, am2=rowmap(lda2) if A2 is not None else None, K2=K2, B2=B2, ldb2=K2 if B2 is not None else 0,
                 a2_ngroup=ngroup,
                 C16=C16, c16m=rowmap(ldc or lin.N) if C16 is not None else None,
                 G16=G16, g16m=rowmap(lin.N) if G16 is not None else None,
                 R32=R32, H32=H32, h32m=rowmap(lin.N) if H32 is not None else None,
                 flags=(GELU_FWD if C16 is not None else NS_GEMM_GELU) if gelu else 0)   # (no pre-activation output: inference, plain GELU)

    def _dgrad(self, dy16, M, lin: _Lin, out16, *, ldy=None, P16=None, A2=None, lda2=0, K2=0, B2=None,
               R32=None, H32=None, drop=False):
        """dx = dy * W  (+ du * A for LoRA), optional gelu' epilogue or fp32 accumulate."""
        self._gemm(A=dy16, am=rowmap(ldy or lin.N), K=lin.N, B=lin.wt, ldb=lin.N, M=M, N=lin.K,
                 A2=A2, am2=rowmap(lda2) if A2 is not None else None, K2=K2, B2=B2, ldb2=K2 if B2 is not None else 0,
                 C16=out16, c16m=rowmap(lin.K) if out16 is not None else None,
                 P16=P16, p16m=rowmap(lin.K) if P16 is not None else None,
                 R32=R32, H32=H32, h32m=rowmap(lin.K) if H32 is not None else None,
                 flags=NS_GEMM_MUL_P16 if P16 is not None else 0,
                 drop_p=self._drop_p() if drop else 0.0, drop_seed=self._cur_seed if drop else 0)

    def _ad_fwd(self, ad, x16, rows, lin: _Lin, u16, seed, **kw):
        """adapted linear, forward: u = drop(x) A^T / keep (all groups stacked), y = x W^T + b + u_g sB_g^T"""
        r, G, dp = self.r, ad["G"], self._drop_p()
        self._gemm(A=x16, am=rowmap(ad["kin"]), K=ad["kin"], B=ad["A"], ldb=ad["kin"], M=rows, N=G * r, C16=u16,
                 c16m=rowmap(G * r), flags=NS_GEMM_DROP_A if dp > 0 else 0, alpha=self._drop_inv(), drop_p=dp, drop_seed=seed)
        self._lin(x16, rows, lin, A2=u16, lda2=G * r, K2=r, B2=ad["sB"], ngroup=ad["nout"] if G > 1 else 0, **kw)

    def _ad_bwd(self, ad, dy16, rows, lin: _Lin, x16, u16, du16, seed, out16, **kw):
        """adapted linear, backward: du_g = dy_g sB_g / keep, dB_g = s dy_g^T u_g, dA = du^T drop(x),
        dx = dy W + mask * (du A)"""
        r, G, nout, kin = self.r, ad["G"], ad["nout"], ad["kin"]
        ldy = G * nout
        for g in range(G):
            self._gemm(A=(dy16, g * nout), am=rowmap(ldy), K=nout, B=ad["sBT"][g], ldb=nout, M=rows, N=r, C16=(du16, g * r),
                     c16m=rowmap(G * r), alpha=self._drop_inv())
            self._wgrad_b((dy16, g * nout), ldy, (u16, g * r), G * r, rows, nout, ad["projs"][g], ad["alpha"][g],
                          ename=ad["site"] + ".lora_E", eoff=g * r)
        self._with_seed(seed, lambda: self._wgrad(du16, G * r, x16, kin, rows, G * r, kin, ad["site"] + ".lora_A", drop=True))
        self._with_seed(seed, lambda: self._dgrad(dy16, rows, lin, out16, A2=du16, lda2=G * r, K2=G * r, B2=ad["AT"],
                                                   drop=True, **kw))

    def _drop_p(self):
        return self.lora.dropout if (self.lora and self.training_mode) else 0.0

    def _drop_inv(self):
        """1/(1-p) of the survivors, p quantised like the kernels' byte mask (thr8/256).  The kernels apply the mask
        only; this factor rides in the alpha of the GEMM that produces the LoRA bottleneck (u forward, du backward)."""
        p = self._drop_p()
        return 256.0 / (256.0 - int(p * 256.0 + 0.5)) if p > 0 else 1.0

    def _wgrad(self, dy16, ldy, x16, ldx, Mred, No, Ko, gname, alpha=1.0, goff=0, ldc=None, drop=False,
               am=None, bm=None, bias_gname=None):
        """bias_gname: also accumulate the column sums of dy (a conv's bias gradient) from the same pass"""
        gptr = self.G.data_ptr() + 4 * (self.seg_off[gname][0] + goff)
        tiles = ((No + 127) // 128) * ((Ko + 127) // 128)
        # blocks in flight: ~1.5 per CU for the 128 x 32 tiles of dB (more splits only add atomics: 23 us at 384 blocks,
        # 30 us at 768 for N = 512), ~3 per CU for the others (tools/probe/tn_splits.py)
        target = 384 if (Ko <= 96 or 32 < No <= 128) else _TN_TARGET
        # (short reductions -- the decoder's adapters under --ft_full, 2816 rows -- split down to one 64-row step per workgroup:
        # with 256-row ranges a 32 x 512 gradient ran on 44 workgroups of four dependent steps each, 29 us)
        splits = max(1, min(Mred // (256 if Mred >= 16384 else 64), (target + tiles - 1) // tiles))
        bptr = self.G.data_ptr() + 4 * self.seg_off[bias_gname][0] if bias_gname else None
        self._gemm(A=dy16, am=am or rowmap(ldy), K=Mred, B=x16, bm=bm or rowmap(ldx), M=No, N=Ko, C32=gptr,
                 ldc32=ldc or Ko, flags=NS_GEMM_TN | NS_GEMM_ATOMIC32 | (NS_GEMM_COLSUM_A if bias_gname else 0), splits=splits,
                 alpha=alpha, H32=bptr, drop_p=self._drop_p() if drop else 0.0, drop_seed=self._cur_seed if drop else 0)

    # ------------------------------------------------------------------ forward
    def _mark(self, name):
        """bench.py's section timing (encoder-only forward / backward): HIP events on the launch stream, only when
        `section_events` is a dict"""
        ev = getattr(self, "section_events", None)
        if ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            ev[name] = e

    def encode(self, x32: torch.Tensor, b: dict, train: bool):
        """MEG signal (B, ch, T) fp32 -> encoder states enc16 (B*S, d) fp16."""
        self._mark("enc_fwd_begin")
        dims, d, f, r, H = self.dims, self.dims.d, self.dims.ffn, self.r, self.dims.heads
        B, T, S, Cp = b["B"], dims.T, dims.src_pos, dims.ch_pad
        M = B * S
        if isinstance(x32, (PackedSignal, _ResidentSignal)):       # on-GPU feed: the batch arrives already packed (feed.py)
            assert x32.shape == (B, dims.ch, T) and x32.xin.shape == (B, T + 2, Cp)
            xin = x32.acquire().xin
        else:
            assert x32.shape == (B, dims.ch, T) and x32.dtype == F32 and x32.is_contiguous()
            xin = b["xin"]
            ops.signal_pack(x32, xin, B, dims.ch, T, Cp)
        b["xin_cur"] = xin                      # the conv weight gradients read it again in backward
        c2 = self.conv_ops["conv2"]
        pb = lambda n: self.pview(f"model.encoder.{n}.bias")  # noqa: E731
        T2 = T // 2
        gelu_f = GELU_FWD if train else NS_GEMM_GELU      # inference keeps no gelu' image (the C16 outputs below are None then)
        if self.frontend == "base":
            c0, c1 = self.conv_ops["conv1.0"], self.conv_ops["conv1.2"]
            # conv1.0 (k3,s1) + GELU  -> pre0 (plain), g0 (halo layout)
            self._gemm(A=xin, am=rowmap(Cp, T, (T + 2) * Cp), K=3 * Cp, B=c0["w"], ldb=3 * Cp, M=B * T, N=d, k_alg=3 * dims.ch,
                     bias=pb("conv1.0"), C16=b["pre0"] if train else None, c16m=rowmap(d), G16=(b["g0"], d),
                     g16m=rowmap(d, T, (T + 2) * d), flags=gelu_f)
            # conv1.2 (k3,s2) + the encoder's outer GELU
            self._gemm(A=b["g0"], am=rowmap(2 * d, T2, (T + 2) * d), K=3 * d, B=c1["w"], ldb=3 * d, M=B * T2, N=d,
                     bias=pb("conv1.2"), C16=b["pre1"] if train else None, c16m=rowmap(d), G16=(b["g1"], d),
                     g16m=rowmap(d, T2, (T2 + 2) * d), flags=gelu_f)
        else:
            # 'replace': one stride-2 conv over the packed signal + the encoder's outer GELU
            cr = self.conv_ops["conv1"]
            self._gemm(A=xin, am=rowmap(2 * Cp, T2, (T + 2) * Cp), K=3 * Cp, B=cr["w"], ldb=3 * Cp, M=B * T2, N=d,
                     bias=pb("conv1"), C16=b["pre1"] if train else None, c16m=rowmap(d), G16=(b["g1"], d),
                     g16m=rowmap(d, T2, (T2 + 2) * d), flags=gelu_f)
        # encoder.conv2 (k3,s2) + GELU + positions -> fp32 residual stream
        h = b["h"]
        self._gemm(A=b["g1"], am=rowmap(2 * d, S, (T2 + 2) * d), K=3 * d, B=c2["w"], ldb=3 * d, M=M, N=d,
                 bias=pb("conv2"), C16=b["pre2"] if train else None, c16m=rowmap(d), H32=h[0], h32m=rowmap(d), pos=self.enc_pos,
                 pos_rows=S, flags=gelu_f)
        dp = self._drop_p()
        rank = r

        def lin_ln(x16, lin, R32, H32, ln, xout, st, A2=None, lda2=0, K2=0, B2=None):
            """residual Linear + the LayerNorm that reads its result: one launch (ns_gemm_ln, bitwise the pair's results) where built"""
            # (only where ns_gemm itself takes the phase-interleaved 256 x 256 kernel -- >= 192 tiles --, whose products ns_gemm_ln repeats
            # bit for bit: below that the two launches run the 128 x 128 ring kernel and the fused form would differ in the last bit)
            if self.use_rowln and (self.use_rowln == 1 or lin.K == lin.N) and ((M + 255) // 256) * ((lin.N + 255) // 256) >= 192 and \
                    ops.gemm_ln_supported(M, lin.N, lin.K, K2):
                ops.gemm_ln(A=x16, am=rowmap(lin.K), K=lin.K, B=lin.w, ldb=lin.K, M=M, N=lin.N, bias=lin.bias, A2=A2,
                            am2=rowmap(lda2) if A2 is not None else None, K2=K2, B2=B2, ldb2=K2 if B2 is not None else 0,
                            R32=R32, H32=H32, h32m=rowmap(lin.N), gamma=ln[0], beta=ln[1], x16=xout, ldx=d, mean=st[0], rstd=st[1])
                return
            self._lin(x16, M, lin, R32=R32, H32=H32, A2=A2, lda2=lda2, K2=K2, B2=B2)
            ops.layernorm_fwd(H32, *ln, xout, *st, M, d)

        ne = dims.enc_layers
        ln1_done = final_done = False
        for i, Lw in enumerate(self.enc):
            r = rank if i < self.n_lora else 0     # --fine_tune_layers: only the first n_lora layers carry adapters
            j = i if train else 0
            if train:
                hin, hmid, hout = h[2 * i], h[2 * i + 1], h[2 * i + 2]
            else:
                hin, hmid, hout = h[0], h[1], h[0]
            lo = self.lora_ops[i] if r else None
            seed = self._layer_seed(i)
            if not ln1_done:      # (else the previous layer's fc2 launch normalised this layer's input already)
                ops.layernorm_fwd(hin, *Lw["ln1"], b["x1"][j], *b["st1"][j], M, d)
            ln1_done = False
            if r:
                self._gemm(A=b["x1"][j], am=rowmap(d), K=d, B=lo["Aqkv"], ldb=d, M=M, N=3 * r, C16=b["uqkv"][j],
                         c16m=rowmap(3 * r), flags=NS_GEMM_DROP_A if dp > 0 else 0, alpha=self._drop_inv(), drop_p=dp, drop_seed=seed)
                self._lin(b["x1"][j], M, Lw["qkv"], C16=b["qkv"][j], A2=b["uqkv"][j], lda2=3 * r, K2=r, B2=lo["sBqkv"],
                          ngroup=d)
            else:
                self._lin(b["x1"][j], M, Lw["qkv"], C16=b["qkv"][j])
            qkv = b["qkv"][j]
            ops.attn_fwd(Q=qkv, K=(qkv, d), V=(qkv, 2 * d), O=b["ao"][j], B=B, H=H, Lq=S, Lk=S, ldq=3 * d, ldk=3 * d,
                         ldv=3 * d, ldo=d, causal=False, LSE=b["lse"][j] if train else None)
            ad = dict(A2=b["uo"][j], lda2=r, K2=r, B2=lo["out_sB"]) if r else {}
            if r:
                self._gemm(A=b["ao"][j], am=rowmap(d), K=d, B=lo["out_A"], ldb=d, M=M, N=r, C16=b["uo"][j], c16m=rowmap(r),
                         flags=NS_GEMM_DROP_A if dp > 0 else 0, alpha=self._drop_inv(), drop_p=dp, drop_seed=seed + 1)
            lin_ln(b["ao"][j], Lw["out"], hin, hmid, Lw["ln2"], b["x2"][j], b["st2"][j], **ad)
            # fc2's result feeds the NEXT layer's first LayerNorm (or the encoder's final one)
            if i + 1 < ne:
                jn = i + 1 if train else 0
                nxt = (self.enc[i + 1]["ln1"], b["x1"][jn], b["st1"][jn])
            else:
                nxt = (self.enc_ln, b["enc16"], b["enc_st"])

            def fc2(**ad2):
                nonlocal ln1_done, final_done
                lin_ln(b["gf"][j], Lw["fc2"], hmid, hout, *nxt, **ad2)
                ln1_done, final_done = i + 1 < ne, i + 1 == ne
            if r:
                self._gemm(A=b["x2"][j], am=rowmap(d), K=d, B=lo["fc1_A"], ldb=d, M=M, N=r, C16=b["u1"][j], c16m=rowmap(r),
                         flags=NS_GEMM_DROP_A if dp > 0 else 0, alpha=self._drop_inv(), drop_p=dp, drop_seed=seed + 2)
                side = train and "u2_slabs" in b and not self.no_side_u2
                sk = dict(side_B=lo["fc2_A"], side_ldb=f, side_n=r, side_out=b["u2_slabs"], side_drop_p=dp,
                          side_drop_seed=seed + 3) if side else {}
                self._lin(b["x2"][j], M, Lw["fc1"], C16=b["pre_f"][j] if train else None, G16=b["gf"][j], gelu=True, A2=b["u1"][j], lda2=r,
                          K2=r, B2=lo["fc1_sB"], **sk)
                if side:   # u2 = drop(gf) A^T / keep from the slabs the GELU epilogue left (no second pass over gf)
                    ops.gemm_side_reduce(b["u2_slabs"], f // 256, M, self._drop_inv(), b["u2"][j], r)
                else:
                    self._gemm(A=b["gf"][j], am=rowmap(f), K=f, B=lo["fc2_A"], ldb=f, M=M, N=r, C16=b["u2"][j], c16m=rowmap(r),
                             flags=NS_GEMM_DROP_A if dp > 0 else 0, alpha=self._drop_inv(), drop_p=dp, drop_seed=seed + 3)
                fc2(A2=b["u2"][j], lda2=r, K2=r, B2=lo["fc2_sB"])
            else:
                self._lin(b["x2"][j], M, Lw["fc1"], C16=b["pre_f"][j] if train else None, G16=b["gf"][j], gelu=True)
                fc2()
        r = rank
        hlast = h[2 * dims.enc_layers] if train else h[0]
        b["h_last"] = hlast
        if not final_done:
            ops.layernorm_fwd(hlast, *self.enc_ln, b["enc16"], *b["enc_st"], M, d)
        self._mark("enc_fwd_end")
        return b["enc16"]

    def _layer_seed(self, i):
        return (self._cur_seed + 16 * i) & 0x7FFFFFFF

    def decode_train(self, dec_ids: torch.Tensor, b: dict, train: bool):
        """Teacher-forced decoder over the whole label sequence -> logits (B*L, Vp) fp16."""
        dims, d, f, H = self.dims, self.dims.d, self.dims.ffn, self.dims.heads
        B, L, S = b["B"], b["L"], dims.src_pos
        M, ML = B * S, B * L
        hd = b["hd"]
        ops.embed_pos(dec_ids, self.E32, self.dec_pos, hd[0], ML, L, d)
        enc16 = b["enc16"]
        kld = b["kv_ld"]
        batched = "kv_all" in b
        if batched:     # every layer's cross K | V in one GEMM over the encoder rows
            self._lin(enc16, M, self.ckv_all, C16=b["kv_all"])
        for i, Lw in enumerate(self.dec):
            j = i if train else 0
            if train:
                h0, h1, h2, h3 = hd[3 * i], hd[3 * i + 1], hd[3 * i + 2], hd[3 * i + 3]
            else:
                h0, h1, h2, h3 = hd[0], hd[1], hd[0], hd[1]
                # eval ping-pong: after the layer the stream must be back in hd[0]
            ads = self.dec_ads[i] if self.dec_lora else None
            sd0 = self._layer_seed(dims.enc_layers + i)

            def lin(key, x16, rows, site_no, **kw):
                if ads is None:
                    self._lin(x16, rows, Lw[key], **kw)
                else:
                    self._ad_fwd(ads[key], x16, rows, Lw[key], b["ud_" + key][j], sd0 + site_no, **kw)
            ops.layernorm_fwd(h0, *Lw["ln1"], b["xs"][j], *b["st_s"][j], ML, d)
            lin("qkv", b["xs"][j], ML, 0, C16=b["qkv_s"][j])
            q = b["qkv_s"][j]
            ops.attn_fwd(Q=q, K=(q, d), V=(q, 2 * d), O=b["ao_s"][j], B=B, H=H, Lq=L, Lk=L, ldq=3 * d, ldk=3 * d,
                         ldv=3 * d, ldo=d, causal=True, LSE=b["lse_s"][j])
            lin("out", b["ao_s"][j], ML, 1, R32=h0, H32=h1)
            ops.layernorm_fwd(h1, *Lw["ln2"], b["xc"][j], *b["st_c"][j], ML, d)
            lin("cq", b["xc"][j], ML, 2, C16=b["q_c"][j])
            kvt, ko = b["kv_c"][j]
            if not batched:
                lin("ckv", enc16, M, 3, C16=kvt)
            ops.attn_fwd(Q=b["q_c"][j], K=(kvt, ko), V=(kvt, ko + d), O=b["ao_c"][j], B=B, H=H, Lq=L, Lk=S, ldq=d, ldk=kld,
                         ldv=kld, ldo=d, causal=False, LSE=b["lse_c"][j])
            lin("cout", b["ao_c"][j], ML, 4, R32=h1, H32=h2)
            ops.layernorm_fwd(h2, *Lw["ln3"], b["xm"][j], *b["st_m"][j], ML, d)
            lin("fc1", b["xm"][j], ML, 5, C16=b["pre_fd"][j], G16=b["gfd"][j], gelu=True)
            lin("fc2", b["gfd"][j], ML, 6, R32=h2, H32=h3)
            if not train and h3 is not hd[0]:
                hd[0].copy_(h3)
        hl = hd[3 * dims.dec_layers] if train else hd[0]
        b["hd_last"] = hl
        ops.layernorm_fwd(hl, *self.dec_ln, b["xd"], *b["st_d"], ML, d)
        Vp = dims.vocab_pad
        self._gemm(A=b["xd"], am=rowmap(d), K=d, B=self.E16, ldb=d, M=ML, N=Vp, C16=b["logits"], c16m=rowmap(Vp))
        return b["logits"]

    @staticmethod
    def shift_tokens_right(labels: torch.Tensor, pad_id: int, start_id: int) -> torch.Tensor:
        """HF:modeling_whisper.py:68-81."""
        out = labels.new_zeros(labels.shape)
        out[:, 1:] = labels[:, :-1]
        out[:, 0] = start_id
        out.masked_fill_(out == -100, pad_id)
        return out

    def forward(self, x32: torch.Tensor, labels: torch.Tensor | None = None, decoder_input_ids=None,
                train: bool = False, compute_grad: bool = False):
        """One forward pass.  Returns (loss_dev or None, logits view (B, L, V))."""
        dims = self.dims
        B = x32.shape[0]
        if decoder_input_ids is None:
            decoder_input_ids = self.shift_tokens_right(labels, dims.pad_id, dims.start_id)
        L = decoder_input_ids.shape[1]
        self.training_mode = train
        self._cur_seed = (self.drop_seed * 2654435761 + 12345) & 0x7FFFFFFF
        b = self._alloc(B, L, train)
        self._b = b
        self.encode(x32, b, train)
        logits = self.decode_train(decoder_input_ids.contiguous(), b, train)
        loss = None
        if labels is not None:
            lab = labels.contiguous().view(-1)
            ops.cross_entropy(logits, lab, B * L, dims.vocab, dims.vocab_pad, b["row_loss"],
                              logits if compute_grad else None, self.nvalid_dev,
                              self.loss_scale_dev if compute_grad else None, self.loss_dev)
            loss = self.loss_dev
            if self.adalora and compute_grad:
                # AdaLoRA's orthogonality regulariser (value into reg_dev, gradient into G, which the caller zeroed)
                ops.zero_(self.reg_dev)
                ops.orth_reg(self._orth_table, self._n_orth, self.lora.orth_reg_weight / self._n_orth,
                             self.loss_scale_dev, self.reg_dev)
                torch.add(self.loss_dev, self.reg_dev, out=self.total_loss_dev)     # static buffer: graph replays write it too
                loss = self.total_loss_dev
        return loss, logits.view(B, L, dims.vocab_pad)[:, :, :dims.vocab]

    # ------------------------------------------------------------------ backward
    def backward(self, on_ready=None):
        """Backward of the last forward(train=True, compute_grad=True); logits buffer holds dlogits.
        Fills self.G (which must have been zeroed).  on_ready(lo, hi) is called when G[lo:hi] is final
        (chunks in backward-completion order) so the caller can start the RCCL all-reduce early."""
        b = self._b
        dims, d, f, r, H = self.dims, self.dims.d, self.dims.ffn, self.r, self.dims.heads
        B, L, S, T = b["B"], b["L"], dims.src_pos, dims.T
        M, ML, Vp = B * S, B * L, dims.vocab_pad
        hd = b["hd"]
        # ---- LM head + final decoder LN
        # K = the padded vocabulary (51 968) against 22 output tiles: split K over ~1000 workgroups, fp32 atomics into
        # a zeroed buffer that the LayerNorm backward reads as its fp32 input
        tiles = ((ML + 127) // 128) * ((d + 127) // 128)
        splits = max(1, min(64, Vp // 2048, -(-1024 // tiles)))
        if splits > 1:
            ops.zero_(b["ddx32"])
            self._gemm(A=b["logits"], am=rowmap(Vp), K=Vp, B=self.E16T, ldb=Vp, M=ML, N=d, C32=b["ddx32"], ldc32=d, splits=splits)
            ops.layernorm_bwd(b["ddx32"], True, hd[3 * dims.dec_layers], *b["st_d"], self.dec_ln[0], None, b["ddh32"],
                              b["ddh16"], ML, d)
        else:
            self._gemm(A=b["logits"], am=rowmap(Vp), K=Vp, B=self.E16T, ldb=Vp, M=ML, N=d, C16=b["ddx16"], c16m=rowmap(d))
            ops.layernorm_bwd(b["ddx16"], False, hd[3 * dims.dec_layers], *b["st_d"], self.dec_ln[0], None, b["ddh32"],
                              b["ddh16"], ML, d)
        first_enc = True
        for i in reversed(range(dims.dec_layers)):
            Lw = self.dec[i]
            h0, h1, h2 = hd[3 * i], hd[3 * i + 1], hd[3 * i + 2]
            ads = self.dec_ads[i] if self.dec_lora else None
            sd0 = self._layer_seed(dims.enc_layers + i)

            def dgrad(key, dy16, rows, x16, site_no, out16, **kw):
                if ads is None:
                    self._dgrad(dy16, rows, Lw[key], out16, **kw)
                else:
                    self._ad_bwd(ads[key], dy16, rows, Lw[key], x16, b["ud_" + key][i],
                                 b["ddu_kv"] if key == "ckv" else b["ddu"], sd0 + site_no, out16, **kw)
            # MLP
            dgrad("fc2", b["ddh16"], ML, b["gfd"][i], 6, b["ddpre_f"], P16=b["pre_fd"][i])
            dgrad("fc1", b["ddpre_f"], ML, b["xm"][i], 5, b["ddx16"])
            ops.layernorm_bwd(b["ddx16"], False, h2, *b["st_m"][i], Lw["ln3"][0], b["ddh32"], b["ddh32"], b["ddh16"], ML, d)
            # cross attention
            dgrad("cout", b["ddh16"], ML, b["ao_c"][i], 4, b["ddao"])
            (kvt, ko), (dkt, dko), kld = b["kv_c"][i], b["dkv_c"][i], b["kv_ld"]
            ops.attn_bwd(Q=b["q_c"][i], K=(kvt, ko), V=(kvt, ko + d), O=b["ao_c"][i], B=B, H=H, Lq=L, Lk=S, ldq=d, ldk=kld,
                         ldv=kld, ldo=d, causal=False, LSE=b["lse_c"][i], dO=b["ddao"], dQ=b["ddq_c"], dK=(dkt, dko),
                         dV=(dkt, dko + d), Delta=b["ddelta"], lddo=d, lddq=d, lddk=kld, lddv=kld, workspace=b["attn_ws_c"])
            dgrad("cq", b["ddq_c"], ML, b["xc"][i], 2, b["ddx16"])
            ops.layernorm_bwd(b["ddx16"], False, h1, *b["st_c"][i], Lw["ln2"][0], b["ddh32"], b["ddh32"], b["ddh16"], ML, d)
            # encoder-state gradient accumulates in fp32 across the decoder layers
            if "dkv_all" not in b:
                dgrad("ckv", dkt, M, b["enc16"], 3, None, R32=None if first_enc else b["denc32"], H32=b["denc32"])
            first_enc = False
            # causal self attention
            dgrad("out", b["ddh16"], ML, b["ao_s"][i], 1, b["ddao"])
            q = b["qkv_s"][i]
            dq = b["ddqkv"]
            ops.attn_bwd(Q=q, K=(q, d), V=(q, 2 * d), O=b["ao_s"][i], B=B, H=H, Lq=L, Lk=L, ldq=3 * d, ldk=3 * d,
                         ldv=3 * d, ldo=d, causal=True, LSE=b["lse_s"][i], dO=b["ddao"], dQ=dq, dK=(dq, d),
                         dV=(dq, 2 * d), Delta=b["ddelta"], lddo=d, lddq=3 * d, lddk=3 * d, lddv=3 * d)
            dgrad("qkv", dq, ML, b["xs"][i], 0, b["ddx16"])
            ops.layernorm_bwd(b["ddx16"], False, h0, *b["st_s"][i], Lw["ln1"][0], b["ddh32"], b["ddh32"], b["ddh16"], ML, d)
        if "dkv_all" in b:      # the encoder-state gradient of all decoder layers' K | V projections: one GEMM over K = layers x 2d
            self._dgrad(b["dkv_all"], M, self.ckv_all, None, H32=b["denc32"])
        # ---- encoder
        self._mark("enc_bwd_begin")
        h = b["h"]
        ops.layernorm_bwd(b["denc32"], True, h[2 * dims.enc_layers], *b["enc_st"], self.enc_ln[0], None, b["dh32"],
                          b["dh16"], M, d)
        sc = self.lora.scale if r else 0.0
        qs = 64 ** -0.5
        rank, nl = r, self.n_lora
        for i in reversed(range(dims.enc_layers)):
            r = rank if i < nl else 0
            Lw = self.enc[i]
            p = f"model.encoder.layers.{i}."
            hin, hmid = h[2 * i], h[2 * i + 1]
            lo = self.lora_ops[i] if r else None
            seed = self._layer_seed(i)
            dy = b["dh16"]
            if r:
                # fc2: du = dy*sB / keep ; dB = s*dy^T u ; dA = du^T mask(gf) ; dgf = dy*W + mask * (du*A)
                # (the kernels apply the dropout MASK only: 1/keep rides in du's alpha, forward in u's alpha)
                self._lora_du_db(dy, d, d, M, b["u2"][i], b["du"], [lo["fc2_sBT"]], [p + "fc2"], [sc], ada_slot="fc2")
                self._with_seed(seed + 3, lambda: self._wgrad(b["du"], r, b["gf"][i], f, M, r, f, p + "fc2.lora_A", drop=True))
                self._with_seed(seed + 3, lambda: self._dgrad(dy, M, Lw["fc2"], b["dpre_f"], P16=b["pre_f"][i], A2=b["du"],
                                                               lda2=r, K2=r, B2=lo["fc2_AT"], drop=True))
                # fc1
                dpf = b["dpre_f"]
                self._lora_du_db(dpf, f, f, M, b["u1"][i], b["du"], [lo["fc1_sBT"]], [p + "fc1"], [sc], ada_slot="fc1")
                self._with_seed(seed + 2, lambda: self._wgrad(b["du"], r, b["x2"][i], d, M, r, d, p + "fc1.lora_A", drop=True))
                self._with_seed(seed + 2, lambda: self._dgrad(dpf, M, Lw["fc1"], b["dx16"], A2=b["du"], lda2=r, K2=r,
                                                               B2=lo["fc1_AT"], drop=True))
            else:
                self._dgrad(dy, M, Lw["fc2"], b["dpre_f"], P16=b["pre_f"][i])
                self._dgrad(b["dpre_f"], M, Lw["fc1"], b["dx16"])
            ops.layernorm_bwd(b["dx16"], False, hmid, *b["st2"][i], Lw["ln2"][0], b["dh32"], b["dh32"], b["dh16"], M, d)
            dy = b["dh16"]
            if r:
                self._lora_du_db(dy, d, d, M, b["uo"][i], b["du"], [lo["out_sBT"]], [p + "self_attn.out_proj"], [sc], ada_slot="out")
                self._with_seed(seed + 1, lambda: self._wgrad(b["du"], r, b["ao"][i], d, M, r, d,
                                                               p + "self_attn.out_proj.lora_A", drop=True))
                self._with_seed(seed + 1, lambda: self._dgrad(dy, M, Lw["out"], b["dao"], A2=b["du"], lda2=r, K2=r,
                                                               B2=lo["out_AT"], drop=True))
            else:
                self._dgrad(dy, M, Lw["out"], b["dao"])
            qkv, dqkv = b["qkv"][i], b["dqkv"]
            ops.attn_bwd(Q=qkv, K=(qkv, d), V=(qkv, 2 * d), O=b["ao"][i], B=B, H=H, Lq=S, Lk=S, ldq=3 * d, ldk=3 * d,
                         ldv=3 * d, ldo=d, causal=False, LSE=b["lse"][i], dO=b["dao"], dQ=dqkv, dK=(dqkv, d),
                         dV=(dqkv, 2 * d), Delta=b["delta"], lddo=d, lddq=3 * d, lddk=3 * d, lddv=3 * d, workspace=b["attn_ws"])
            if r:
                self._lora_du_db(dqkv, 3 * d, d, M, b["uqkv"][i], b["du3"], [lo["sBqT"], lo["sBkT"], lo["sBvT"]],
                                 [p + f"self_attn.{nm}" for nm in ("q_proj", "k_proj", "v_proj")], [sc * qs, sc, sc],
                                 enames=[(p + "self_attn.qkv.lora_E", j * r) for j in range(3)], ada_slot="qkv")
                self._with_seed(seed, lambda: self._wgrad(b["du3"], 3 * r, b["x1"][i], d, M, 3 * r, d,
                                                           p + "self_attn.qkv.lora_A", drop=True))
                self._with_seed(seed, lambda: self._dgrad(dqkv, M, Lw["qkv"], b["dx16"], A2=b["du3"], lda2=3 * r, K2=3 * r,
                                                           B2=lo["AqkvT"], drop=True))
            else:
                self._dgrad(dqkv, M, Lw["qkv"], b["dx16"])
            ops.layernorm_bwd(b["dx16"], False, hin, *b["st1"][i], Lw["ln1"][0], b["dh32"], b["dh32"], b["dh16"], M, d)
            if r and self.adalora:
                ops.adalora_fold_jobs(*self._fold_tables[i])     # dB / dE of this layer's six projections, scratch cleared
            if on_ready is not None and r and i in (nl // 2, 0):
                # adapter gradients of layers [i, hi_l] are final: two chunks (upper half, lower half of the adapted layers)
                hi_l = nl - 1 if (i == nl // 2 and i != 0) or nl // 2 == 0 else nl // 2 - 1
                lo_off = self.seg_off[f"model.encoder.layers.{hi_l}.self_attn.qkv.lora_A"][0]
                if hi_l == nl - 1:
                    lo_off = 0      # --ft_full: the decoder adapters (front of the buffer) were final before the encoder started
                end = self.lora_end if i == 0 else self.seg_off[f"model.encoder.layers.{i - 1}.self_attn.qkv.lora_A"][0]
                on_ready(lo_off, end)
        r = rank
        # ---- conv stem (all three convs are trainable: modules_to_save, finetune.py:202)
        if self.train_convs:
            self._stem_backward(b)
        self._mark("enc_bwd_end")
        if on_ready is not None:
            on_ready(self.lora_end, self.n_train)

    def _wgrad_b(self, dy16, ldy, u16, ldu, Mred, N, key, s, ename=None, eoff=0):
        """LoRA-B weight gradient dB = s * dY^T u.  AdaLoRA: the GEMM yields the gradient of the folded operand
        s*B*diag(E); ns_adalora_fold_grads turns it into dB and dE."""
        r = self.r
        if not self.adalora:
            self._wgrad(dy16, ldy, u16, ldu, Mred, N, r, key + ".lora_B", alpha=s)
            return
        tmp = self._gbf[:N * r]
        ops.zero_(tmp)
        self._gemm(A=dy16, am=rowmap(ldy), K=Mred, B=u16, bm=rowmap(ldu), M=N, N=r, C32=tmp, ldc32=r,
                 flags=NS_GEMM_TN | NS_GEMM_ATOMIC32, splits=max(1, min(Mred // 256, -(-384 // ((N + 127) // 128)))))
        en = ename or key + ".lora_E"
        ops.adalora_fold_grads(tmp, self.pview(key + ".lora_B"), (self.pview(en), eoff), self.gview(key + ".lora_B"),
                               (self.gview(en), eoff), N, r, s)

    def _lora_du_db(self, dy16, ldy, N, M, u16, du16, sBT, keys, alphas, enames=None, ada_slot=None):
        """Backward of the adapter up-projections of ONE site (G = len(keys) column groups of dy: q | k | v, else one):
        du_g = dy_g sB_g / keep and dB_g = s dy_g^T u_g.  One fused pass over dy (ns_lora_bwd_dudb) where the shape is
        built, else the skinny GEMM + weight-gradient GEMM pair.  AdaLoRA: the product lands in a scratch buffer and
        ns_adalora_fold_grads turns the gradient of the folded operand s*B*diag(E) into dB and dE."""
        r, G = self.r, len(keys)
        # (fused vs the two GEMMs at M = 96 000, profiles/r2_b: q | k | v 131 vs 153 us, out / fc2 44 vs 51 us, fc1 165 vs 176 us;
        # its dB partials leave by fp32 atomics, N x r x 4 B per workgroup, which is what keeps the N = 2048 site close)
        if ops.lora_bwd_supported(N, r, G) and not self.no_fused_lora_bwd:
            if self.adalora and ada_slot is not None:
                # the layer's scratch (clear: ns_adalora_fold_jobs zeroes what it folds); folded once per layer by the caller
                o = self._ada_off[ada_slot]
                tmp = self._ada_scr[o:o + G * N * r].view(G, N * r)
                dB = [tmp[g] for g in range(G)]
                al = [1.0] * G
            elif self.adalora:
                tmp = self._gbf3[:G * N * r].view(G, N * r)
                ops.zero_(tmp)
                dB = [tmp[g] for g in range(G)]
                al = [1.0] * G
            else:
                dB = [self.gview(k + ".lora_B") for k in keys]
                al = list(alphas)
            ops.lora_bwd_dudb(dy=dy16, ldy=ldy, u=u16, ldu=G * r, du=du16, lddu=G * r, sBT=sBT, dB=dB, lddb=r, M=M, N=N, r=r,
                              alpha_du=self._drop_inv(), alpha_db=al)
            if self.adalora and ada_slot is None:
                for g, k in enumerate(keys):
                    en, eoff = enames[g] if enames else (k + ".lora_E", 0)
                    ops.adalora_fold_grads(tmp[g], self.pview(k + ".lora_B"), (self.pview(en), eoff), self.gview(k + ".lora_B"),
                                           (self.gview(en), eoff), N, r, alphas[g])
            return
        if G > 1 and not self.adalora and not self.no_fused_lora_bwd and ops.lora_bwd_supported(N, r, 1):
            # a stacked site whose G-group form is not built (whisper-large-v2's q | k | v: three groups of 1280 columns would need 240
            # accumulator registers per thread) runs the fused kernel once per GROUP on that group's columns of dy / u / du: the same bytes
            # of dy in total, three launches instead of three down-projection GEMMs + three weight-gradient GEMMs (round 6: 3 x 77 us
            # against 3 x 57 + 3 x 90 us per layer, profiles/r6_large_v2_kernel_stats.csv)
            for g, k in enumerate(keys):
                ops.lora_bwd_dudb(dy=(dy16, g * N), ldy=ldy, u=(u16, g * r), ldu=G * r, du=(du16, g * r), lddu=G * r, sBT=[sBT[g]],
                                  dB=[self.gview(k + ".lora_B")], lddb=r, M=M, N=N, r=r, alpha_du=self._drop_inv(), alpha_db=[alphas[g]])
            return
        for g, k in enumerate(keys):
            self._gemm(A=(dy16, g * N) if G > 1 else dy16, am=rowmap(ldy), K=N, B=sBT[g], ldb=N, M=M, N=r,
                     C16=(du16, g * r) if G > 1 else du16, c16m=rowmap(G * r), alpha=self._drop_inv())
            en, eoff = enames[g] if enames else (None, 0)
            self._wgrad_b((dy16, g * N) if G > 1 else dy16, ldy, (u16, g * r) if G > 1 else u16, G * r, M, N, k, alphas[g],
                          ename=en, eoff=eoff)

    def _with_seed(self, seed, fn):
        save = self._cur_seed
        self._cur_seed = seed & 0x7FFFFFFF
        try:
            fn()
        finally:
            self._cur_seed = save

    def _stem_backward(self, b):
        dims, d = self.dims, self.dims.d
        B, S, T, Cp = b["B"], dims.src_pos, dims.T, dims.ch_pad
        T2, M = T // 2, B * S
        c2 = self.conv_ops["conv2"]
        gp = lambda name: self.G.data_ptr() + 4 * self.seg_off[name][0]  # noqa: E731
        # d(pre2) = round16(dh) * gelu'(pre2), halo layout (B, S+2, d)
        ops.dgelu_mul(b["dh16"], b["pre2"], (b["dpre2"], d), rowmap(d, S, (S + 2) * d), M, d, pre_is_grad=True)
        dp2 = (b["dpre2"], d)
        hal2 = rowmap(d, S, (S + 2) * d)
        # the bias gradients (column sums of d(pre)) ride on the weight-gradient GEMMs that stream the same rows
        self._wgrad(dp2, 0, b["g1"], 0, M, d, 3 * d, "model.encoder.conv2.wp", am=hal2, bm=rowmap(2 * d, S, (T2 + 2) * d),
                    bias_gname="model.encoder.conv2.bias")
        # dgrad of the stride-2 conv: even output rows (tap 1), odd rows (taps 2|0 over dy[i], dy[i+1])
        # gelu'(pre1) is applied in the epilogue; results land directly in the halo layout of d(pre1)
        ev = rowmap(2 * d, S, T2 * d)
        evh = rowmap(2 * d, S, (T2 + 2) * d)
        self._gemm(A=dp2, am=hal2, K=d, B=c2["we"], ldb=d, M=M, N=d, C16=(b["dpre1"], d), c16m=evh, P16=b["pre1"],
                 p16m=ev, flags=NS_GEMM_MUL_P16)
        self._gemm(A=dp2, am=hal2, K=2 * d, B=c2["wo"], ldb=2 * d, M=M, N=d, C16=(b["dpre1"], 2 * d), c16m=evh,
                 P16=(b["pre1"], d), p16m=ev, flags=NS_GEMM_MUL_P16)
        dp1 = (b["dpre1"], d)
        hal1 = rowmap(d, T2, (T2 + 2) * d)
        if self.frontend == "replace":
            self._wgrad(dp1, 0, b["xin_cur"], 0, B * T2, d, 3 * Cp, "model.encoder.conv1.wp", am=hal1,
                        bm=rowmap(2 * Cp, T2, (T + 2) * Cp), bias_gname="model.encoder.conv1.bias")
            return
        c1 = self.conv_ops["conv1.2"]
        self._wgrad(dp1, 0, b["g0"], 0, B * T2, d, 3 * d, "model.encoder.conv1.2.wp", am=hal1,
                    bm=rowmap(2 * d, T2, (T + 2) * d), bias_gname="model.encoder.conv1.2.bias")
        ev = rowmap(2 * d, T2, T * d)
        self._gemm(A=dp1, am=hal1, K=d, B=c1["we"], ldb=d, M=B * T2, N=d, C16=b["dpre0"], c16m=ev, P16=b["pre0"], p16m=ev,
                 flags=NS_GEMM_MUL_P16)
        self._gemm(A=dp1, am=hal1, K=2 * d, B=c1["wo"], ldb=2 * d, M=B * T2, N=d, C16=(b["dpre0"], d), c16m=ev,
                 P16=(b["pre0"], d), p16m=ev, flags=NS_GEMM_MUL_P16)
        self._wgrad(b["dpre0"], 0, b["xin_cur"], 0, B * T, d, 3 * Cp, "model.encoder.conv1.0.wp", am=rowmap(d, T, T * d),
                    bm=rowmap(Cp, T, (T + 2) * Cp), bias_gname="model.encoder.conv1.0.bias")

    # ------------------------------------------------------------------ optimizer
    def zero_grad(self):
        ops.zero_(self.G)       # a kernel (ns_zero_spans), not a memset node: the step is replayed from hipGraphs

    def optimizer_step(self):
        tc = self.tc
        cfg = AdamWCfg(tc.lr, tc.beta1, tc.beta2, tc.eps, tc.weight_decay, tc.max_grad_norm, tc.warmup_steps,
                       tc.total_steps, tc.growth_factor, tc.backoff_factor, tc.growth_interval)
        ops.grad_norm(self.G, self.n_train, self.norm_ws, self.norm2_dev, self.found_inf_dev)
        ops.adamw_step(self.P, self.G, self.M1, self.M2, self.n_train, cfg, self.step_dev, self.norm2_dev,
                       self.found_inf_dev, self.loss_scale_dev if tc.fp16_scaler else None,
                       self.growth_dev if tc.fp16_scaler else None)
        self.refresh_operands()
        ops.add_i32(self.seed_ctr, 1)   # next step, next dropout masks (device-side: see _init_opt_state)

    def train_step(self, x32, labels, on_ready=None, reduce_fn=None):
        """forward + backward (+ optional gradient reduction) + optimizer; returns the device loss scalar.

        On a GPU the step is captured once per (batch shape, label length, input buffer) in hipGraphs and replayed: a step
        is ~380 launches whose enqueueing costs ~17 ms of Python / ctypes, more than the 15 ms a step may take at
        north_star's 40 % target, and under data parallelism eight such interpreters share one host.  Nothing in a step
        depends on the host: optimizer state, loss scale, schedule and the dropout counter live on the device.  With a
        gradient exchange (`on_ready` / `reduce_fn`: dp.GradReducer) the capture is CUT at every on_ready point, so the
        RCCL chunks are launched eagerly on their side stream between two replays exactly where the eager step launches
        them.  The first step of a shape runs eagerly (lazy allocations), the second is captured."""
        if not self._graph_usable(x32):
            return self._train_step_eager(x32, labels, on_ready, reduce_fn)
        # label lengths vary from batch to batch (the longest transcript of the batch): pad them to a multiple of 16 with the
        # ignore index so that a recipe needs a handful of graphs, not one per length.  Exactly neutral: padded positions are
        # ignored by the loss, lie behind every real position under the causal mask, and the loss is a mean over valid tokens.
        L = labels.shape[1]
        Lp = min((L + self.label_pad - 1) // self.label_pad * self.label_pad, max(L, self.dims.tgt_pos))   # never past the position table
        if Lp != L:
            labels = torch.nn.functional.pad(labels, (0, Lp - L), value=-100)
        packed = isinstance(x32, PackedSignal)
        cut = on_ready is not None or reduce_fn is not None
        # everything a captured launch argument was computed from: shapes, the input buffer, the exchange cuts, and the
        # host-side settings that ride in kernel arguments (optimizer hyper-parameters, dropout rate, the seed base).  A plain
        # fp32 batch is a fresh tensor every step (a new address each time), so it is packed EAGERLY into the static `xin` of
        # its batch size and the capture starts behind the pack: its key holds no input address.  A PackedSignal is one of the
        # feed's few staging slots and is captured by address.
        import dataclasses
        key = (tuple(x32.shape), tuple(labels.shape), x32.xin.data_ptr() if packed else 0, cut, dataclasses.astuple(self.tc),
               self.lora.dropout if self.lora else 0.0, self.drop_seed, self.no_fused_lora_bwd, self.train_convs,
               self.use_rowln)
        g = self._graphs.get(key)
        if g is None:
            warm = (tuple(x32.shape), tuple(labels.shape))
            if warm not in self._graph_warm:
                self._graph_warm.add(warm)
                return self._train_step_eager(x32, labels, on_ready, reduce_fn)
            prev_stream = torch.cuda.current_stream()
            try:
                g = self._capture_step(x32, labels, cut)
                self._graph_captures += 1
            except RuntimeError as e:
                # another host thread outside this package's capture lock (e.g. torch's pin-memory thread) can invalidate a
                # capture on HIP: nothing has executed, so run this step eagerly.
                # torch.cuda.graph.__exit__ ends the capture BEFORE it leaves its side stream: when capture_end raises, the
                # capture stream would stay current (the eager step, the prefetcher's wait_stream / record_stream and the
                # allocator's stream ownership would all move to it) -- put the caller's stream back
                torch.cuda.synchronize()
                torch.cuda.set_stream(prev_stream)
                self._graph_failed_total += 1
                # best effort only: the graveyard keeps the dead graph objects and their pool alive, but a truly
                # invalidated capture has been seen to abort later inside the allocator; a second attempt doubles that exposure
                # and, under DP, a rank that dies mid-run hangs the others in RCCL.  One real failure ends graph use for the run.
                self.use_graph = False
                import warnings
                warnings.warn(f"train_step: hipGraph capture failed ({str(e).splitlines()[0][:120]}); eager steps from here on "
                              "(graphs disabled for this engine)")
                return self._train_step_eager(x32, labels, on_ready, reduce_fn)
            if len(self._graphs) >= self.graph_cache:        # least recently used first (hits move a key to the end)
                self._graphs.pop(next(iter(self._graphs)))
                self._graph_evictions += 1
            self._graphs[key] = g
        else:
            self._graphs[key] = self._graphs.pop(key)        # LRU order
        if packed:
            x32.acquire()            # the copy stream's event: waited for eagerly, never inside a capture
        else:
            # the same contract encode() asserts on the eager path: the pack below reads the raw pointer
            if not (x32.dtype == F32 and x32.is_contiguous() and x32.device == self.dev
                    and tuple(x32.shape) == (x32.shape[0], self.dims.ch, self.dims.T)):
                raise ValueError(f"train_step: the batch must be a contiguous float32 (B, {self.dims.ch}, {self.dims.T}) tensor on "
                                 f"{self.dev} (got {x32.dtype}, {tuple(x32.shape)}, {x32.device}, contiguous={x32.is_contiguous()})")
            ops.signal_pack(x32, g["b"]["xin"], x32.shape[0], self.dims.ch, self.dims.T, self.dims.ch_pad)
        g["labels"].copy_(labels, non_blocking=True)
        self.training_mode = True
        self._b = g["b"]
        self._graph_replays += 1
        for seg, hook in zip(g["segs"], g["hooks"]):
            seg.replay()
            if hook is not None:
                if hook == "reduce":
                    if reduce_fn is not None:
                        reduce_fn()
                elif on_ready is not None:
                    on_ready(*hook)
        return self.loss_dev if not self.adalora else self.total_loss_dev

    def graph_stats(self) -> dict:
        """what bench.py / finetune.py report AFTER a run: a capture that failed and fell back to eager steps must be visible"""
        return {"enabled": bool(self.use_graph), "graphs_cached": len(self._graphs), "captures": self._graph_captures,
                "capture_failures": self._graph_failed_total, "replays": self._graph_replays, "evictions": self._graph_evictions,
                "segments": sorted({len(g["segs"]) for g in self._graphs.values()})}

    def _graph_usable(self, x32):
        if not (self.use_graph and self.dev.type == "cuda"):
            return False
        # the bench's instrumentation (events around every GEMM / section marks) belongs to eager steps
        return ops.GEMM_PROFILE is None and ops.STEP_PROFILE is None and getattr(self, "section_events", None) is None

    def _capture_step(self, x32, labels, cut: bool):
        """Capture one training step.  Nothing executes during capture: the caller replays the segments afterwards.
        cut=True ends a segment at every on_ready point of backward() and before the optimizer (hooks: (lo, hi) of the
        gradient chunk that is final there, "reduce" = the exchange must have completed)."""
        if isinstance(x32, PackedSignal):
            x32.acquire()
        else:
            # a plain fp32 batch: the caller packs it eagerly into the static `xin` of this batch size before every replay
            # (train_step), the captured step starts behind the pack
            x32 = _ResidentSignal(self._alloc_enc(x32.shape[0], True)["xin"], tuple(x32.shape))
        lab = labels.clone()
        torch.cuda.synchronize()
        segs, hooks = [], []
        state = {"g": None, "ctx": None}
        pool = torch.cuda.graph_pool_handle()       # one private pool for all segments of this step

        def begin():
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: the data feed's loader thread issues copies on its own stream meanwhile
            ctx = torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local")
            ctx.__enter__()
            state["g"], state["ctx"] = g, ctx

        def end(hook):
            ctx, state["ctx"] = state["ctx"], None
            ctx.__exit__(None, None, None)          # raises when the capture was invalidated: the caller falls back to eager
            segs.append(state["g"])
            hooks.append(hook)

        def cut_here(lo, hi):
            end((lo, hi))
            begin()
        with GPU_CAPTURE_LOCK:      # no other thread of this package touches the GPU API while the capture is open
            begin()
            try:
                self.zero_grad()
                self.forward(x32, lab, train=True, compute_grad=True)
                self.backward(cut_here if cut else None)
                if cut:
                    end("reduce")
                    begin()
                self.optimizer_step()
                end(None)
            except BaseException:
                if state["ctx"] is not None:        # close the open capture before the error travels on
                    try:
                        state["ctx"].__exit__(None, None, None)
                    except Exception:
                        pass
                # graph objects of a capture that did not end cleanly are never destroyed: torch's ~CUDAGraph of such an object
                # aborted the process on the GPU box (a check that throws inside a destructor) when the frames of this call
                # were released; a failed capture is rare and its objects are small
                self._graph_graveyard.append((state["g"], segs, pool))
                raise
        return {"segs": segs, "hooks": hooks, "labels": lab, "b": self._b}

    def _train_step_eager(self, x32, labels, on_ready=None, reduce_fn=None):
        self.zero_grad()
        loss, _ = self.forward(x32, labels, train=True, compute_grad=True)
        self.backward(on_ready)
        if reduce_fn is not None:
            reduce_fn()
        self.optimizer_step()
        return loss

    def accumulate_step(self, x32, labels, index: int, count: int, on_ready=None, reduce_fn=None):
        """One micro-batch of `count` (finetune.py --gradient_accumulation_steps, :56,235; HF Trainer divides each
        micro-loss by `count`): gradients add up in G, the exchange and the optimizer run with the last one.  Returns
        the micro-batch loss (a clone: the device scalar is overwritten by the next forward)."""
        if index == 0:
            self.zero_grad()
        seed = self.drop_seed
        self.drop_seed = seed * count + index            # a different dropout mask per micro-batch
        loss, _ = self.forward(x32, labels, train=True, compute_grad=True)
        self.drop_seed = seed
        last = index == count - 1
        if last and count > 1:
            # the earlier micro-batches' gradients are already in G: scale the sum once instead of every loss
            self.backward(None)
            self.G.mul_(1.0 / count)
            if on_ready is not None:
                on_ready(0, self.n_train)
        else:
            self.backward(on_ready if last else None)
        out = loss.clone()
        if last:
            if reduce_fn is not None:
                reduce_fn()
            self.optimizer_step()
        return out

    # ------------------------------------------------------------------ export
    def conv_weight(self, name: str) -> torch.Tensor:
        """torch-layout (N, C, 3) view of a packed conv master (name in conv1.0 / conv1.2 / conv2)."""
        d = self.dims.d
        cin, cp = (self.dims.ch, self.dims.ch_pad) if name in ("conv1.0", "conv1") else (d, d)
        return self.pview(f"model.encoder.{name}.wp").view(d, 3, cp).permute(0, 2, 1)[:, :cin, :]

    def conv_weight_grad(self, name: str) -> torch.Tensor:
        d = self.dims.d
        cin, cp = (self.dims.ch, self.dims.ch_pad) if name in ("conv1.0", "conv1") else (d, d)
        return self.gview(f"model.encoder.{name}.wp").view(d, 3, cp).permute(0, 2, 1)[:, :cin, :]
