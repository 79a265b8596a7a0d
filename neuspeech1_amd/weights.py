"""Model dimensions and the documented counter-based weight generator.

No pretrained checkpoints exist offline, so parity is pinned on seeded synthetic
weights.  Every tensor is generated independently from (seed, crc32(name)) with
numpy's Philox bit generator, so the CPU oracle, the HF golden generator and
the GPU engine reproduce the same fp32 values without shipping weight blobs.
Scales are chosen so activations stay O(1) and the softmaxes are not uniform
(std 1/sqrt(fan_in) for matrices), unlike HF's 0.02 init which decodes to
degenerate repeats (SURVEY.md §8c caveat gamma).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, asdict

import numpy as np


@dataclass(frozen=True)
class WhisperDims:
    d: int = 512
    heads: int = 8
    ffn: int = 2048
    enc_layers: int = 6
    dec_layers: int = 6
    vocab: int = 51865
    src_pos: int = 1500          # encoder positions; the MEG window is 4*src_pos samples
    tgt_pos: int = 448
    ch: int = 208                # MEG channels
    pad_id: int = 50257
    bos_id: int = 50257
    eos_id: int = 50257
    start_id: int = 50258        # decoder_start_token_id

    @property
    def T(self) -> int:
        return 4 * self.src_pos

    @property
    def ch_pad(self) -> int:
        """channels of the packed (B, T + 2, Cp) signal image: the next multiple of 16 (the first conv's K = 3 Cp has to be a multiple of 16 for
        the NT GEMM).  Rounds 1-3 padded to 64 -- K = 768 for the algorithmic 624 of the 208-channel first conv (now 624), 960 for 819 at 273
        channels (now 864)"""
        return max(48, (self.ch + 15) // 16 * 16)     # (48: the first conv's weight-gradient kernel wants 3 Cp > 96 -- toy configs only)

    @property
    def vocab_pad(self) -> int:
        return (self.vocab + 127) // 128 * 128

    def to_dict(self):
        return asdict(self)


WHISPER_BASE = WhisperDims()
WHISPER_LARGE_V2 = WhisperDims(d=1280, heads=20, ffn=5120, enc_layers=32, dec_layers=32, ch=273)
# whisper-large-v2 WIDTH (d 1280, 20 heads, ffn 5120, 273 channels) at 2 + 2 layers: the parity fixture of BASELINE configs[4]
LV2W = WhisperDims(d=1280, heads=20, ffn=5120, enc_layers=2, dec_layers=2, ch=273)
# tiny config for fast parity tests (dims kept multiples of what the kernels need)
TINY = WhisperDims(d=256, heads=4, ffn=512, enc_layers=2, dec_layers=2, vocab=1000, src_pos=100, tgt_pos=64, ch=20,
                   pad_id=999, bos_id=999, eos_id=999, start_id=998)


def _gen(name: str, shape, std: float, seed: int, mean: float = 0.0) -> np.ndarray:
    key = (zlib.crc32(name.encode()) + 0x9E3779B9 * (seed + 1)) & 0xFFFFFFFFFFFFFFFF
    rng = np.random.Generator(np.random.Philox(key=key))
    return (mean + std * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)


def make_state_dict(dims: WhisperDims, seed: int = 42, frontend: str = "base", gen=None) -> dict[str, np.ndarray]:
    """HF-named fp32 state dict of Whisper with the MEG front-end installed
    (reference: utils/model_utils.py:9-23 replaces encoder.conv1 by a Sequential whose
    state-dict keys are 0.weight/0.bias/2.weight/2.bias).  `gen(name, shape, std, seed, mean)` replaces the numpy
    generator (bench.py's large-v2 leg draws its 1.5 G random-init parameters on the device: 67 s of host time otherwise)."""
    d, f = dims.d, dims.ffn
    _gen = gen or globals()["_gen"]
    sd: dict[str, np.ndarray] = {}

    def lin(prefix, out_f, in_f, bias=True):
        sd[prefix + ".weight"] = _gen(prefix + ".weight", (out_f, in_f), in_f ** -0.5, seed)
        if bias:
            sd[prefix + ".bias"] = _gen(prefix + ".bias", (out_f,), 0.1, seed)

    def ln(prefix):
        sd[prefix + ".weight"] = _gen(prefix + ".weight", (d,), 0.1, seed, mean=1.0)
        sd[prefix + ".bias"] = _gen(prefix + ".bias", (d,), 0.1, seed)

    def attn(prefix):
        lin(prefix + ".k_proj", d, d, bias=False)
        lin(prefix + ".v_proj", d, d)
        lin(prefix + ".q_proj", d, d)
        lin(prefix + ".out_proj", d, d)

    e = "model.encoder."
    if frontend == "replace":   # projection_module('replace'): one Conv1d(ch, d, k3, s2, p1) (utils/model_utils.py:19-21)
        sd[e + "conv1.weight"] = _gen(e + "conv1.weight", (d, dims.ch, 3), (3 * dims.ch) ** -0.5, seed)
        sd[e + "conv1.bias"] = _gen(e + "conv1.bias", (d,), 0.1, seed)
    else:
        sd[e + "conv1.0.weight"] = _gen(e + "conv1.0.weight", (d, dims.ch, 3), (3 * dims.ch) ** -0.5, seed)
        sd[e + "conv1.0.bias"] = _gen(e + "conv1.0.bias", (d,), 0.1, seed)
        sd[e + "conv1.2.weight"] = _gen(e + "conv1.2.weight", (d, d, 3), (3 * d) ** -0.5, seed)
        sd[e + "conv1.2.bias"] = _gen(e + "conv1.2.bias", (d,), 0.1, seed)
    sd[e + "conv2.weight"] = _gen(e + "conv2.weight", (d, d, 3), (3 * d) ** -0.5, seed)
    sd[e + "conv2.bias"] = _gen(e + "conv2.bias", (d,), 0.1, seed)
    sd[e + "embed_positions.weight"] = _gen(e + "embed_positions.weight", (dims.src_pos, d), 0.3, seed)
    for i in range(dims.enc_layers):
        p = f"{e}layers.{i}."
        attn(p + "self_attn")
        ln(p + "self_attn_layer_norm")
        lin(p + "fc1", f, d)
        lin(p + "fc2", d, f)
        ln(p + "final_layer_norm")
    ln(e + "layer_norm")
    dd = "model.decoder."
    sd[dd + "embed_tokens.weight"] = _gen(dd + "embed_tokens.weight", (dims.vocab, d), d ** -0.5, seed)
    sd[dd + "embed_positions.weight"] = _gen(dd + "embed_positions.weight", (dims.tgt_pos, d), 0.3, seed)
    for i in range(dims.dec_layers):
        p = f"{dd}layers.{i}."
        attn(p + "self_attn")
        ln(p + "self_attn_layer_norm")
        attn(p + "encoder_attn")
        ln(p + "encoder_attn_layer_norm")
        lin(p + "fc1", f, d)
        lin(p + "fc2", d, f)
        ln(p + "final_layer_norm")
    ln(dd + "layer_norm")
    return sd


LORA_SUFFIXES = ("k_proj", "q_proj", "v_proj", "out_proj", "fc1", "fc2")


def make_lora_state(dims: WhisperDims, r: int, seed: int = 7, b_std: float = 0.02, adalora: bool = False,
                    decoder: bool = False) -> dict[str, np.ndarray]:
    """LoRA A/B for every encoder q/k/v/out/fc1/fc2 (finetune.py:189-198) and, with decoder=True, for every decoder
    projection as well (--ft_full, finetune.py:191-192).  PEFT initialises B to zero; tests use a non-zero B (b_std)
    so the side path and its gradients are exercised."""
    d, f = dims.d, dims.ffn
    out = {}
    names = []
    for i in range(dims.enc_layers):
        p = f"model.encoder.layers.{i}."
        names += [(p + (suf if suf in ("fc1", "fc2") else f"self_attn.{suf}"), suf) for suf in LORA_SUFFIXES]
    if decoder:
        for i in range(dims.dec_layers):
            p = f"model.decoder.layers.{i}."
            for suf in LORA_SUFFIXES:
                if suf in ("fc1", "fc2"):
                    names.append((p + suf, suf))
                else:
                    names += [(p + f"self_attn.{suf}", suf), (p + f"encoder_attn.{suf}", suf)]
    for name, suf in names:
        in_f = f if suf == "fc2" else d
        out_f = f if suf == "fc1" else d
        out[name + ".lora_A.weight"] = _gen(name + ".lora_A", (r, in_f), in_f ** -0.5, seed)
        out[name + ".lora_B.weight"] = _gen(name + ".lora_B", (out_f, r), b_std, seed)
        if adalora:   # peft initialises E to zero; tests use a non-zero E so that every gradient path is live
            out[name + ".lora_E.weight"] = _gen(name + ".lora_E", (r, 1), 0.5, seed)
    return out


def synth_batch(dims: WhisperDims, B: int, seed: int = 1234, min_k: int = 8, max_k: int = 40, full_len: bool = True):
    """Synthetic MEG batch of SURVEY.md §8d: x ~ clip(N(0,0.35^2),-1,1) fp32 (B,ch,T);
    labels [start, lang, task, notimestamps] + U{0..specials-1}^k + [eos], padded with -100."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    x = np.clip(0.35 * rng.standard_normal((B, dims.ch, dims.T), dtype=np.float32), -1, 1).astype(np.float32)
    if not full_len:
        for b in range(B):
            n = int(rng.integers(dims.T // 15, dims.T + 1))
            x[b, :, n:] = 0.0
    ks = rng.integers(min_k, max_k + 1, size=B)
    L = int(ks.max()) + 5
    labels = np.full((B, L), -100, dtype=np.int64)
    n_text = dims.vocab - 1608 if dims.vocab > 2000 else dims.vocab - 8   # ids below the special-token block
    prefix = [dims.start_id, dims.start_id + 1, min(dims.start_id + 101, dims.vocab - 1), min(dims.start_id + 105, dims.vocab - 1)]
    if dims.vocab <= 2000:
        prefix = [dims.start_id, dims.vocab - 7, dims.vocab - 6, dims.vocab - 5]
    for b in range(B):
        k = int(ks[b])
        toks = rng.integers(0, n_text, size=k)
        row = prefix + toks.tolist() + [dims.eos_id]
        labels[b, :len(row)] = row
    return x, labels
