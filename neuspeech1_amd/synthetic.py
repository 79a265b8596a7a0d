"""Offline stand-ins so the CLIs run end-to-end without hub files (HF_HUB_OFFLINE, no tokenizer on disk):
a deterministic character-level processor with Whisper's label layout
[<|startoftranscript|>, <|lang|>, <|transcribe|>, <|notimestamps|>, ...text..., <|endoftext|>] and a writer of
synthetic MEG datasets (.npy + .jsonl) in the reference's on-disk format (process_dataset/*.py output)."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .weights import WhisperDims


class _Batch(dict):
    """dict with attribute access, like transformers.BatchEncoding"""
    __getattr__ = dict.__getitem__


class SyntheticTokenizer:
    def __init__(self, dims: WhisperDims):
        self.dims = dims
        v = dims.vocab
        if v > 2000:   # whisper multilingual ids
            self.special = {"<|endoftext|>": 50257, "<|startoftranscript|>": 50258, "<|en|>": 50259, "<|nl|>": 50271,
                            "<|translate|>": 50358, "<|transcribe|>": 50359, "<|nocaptions|>": 50362,
                            "<|notimestamps|>": 50363}
            self.n_text = 50257
        else:
            self.special = {"<|endoftext|>": dims.eos_id, "<|startoftranscript|>": dims.start_id, "<|en|>": v - 7,
                            "<|nl|>": v - 8, "<|translate|>": v - 4, "<|transcribe|>": v - 6, "<|nocaptions|>": v - 3,
                            "<|notimestamps|>": v - 5}
            self.n_text = v - 8
        self.bos_token_id = dims.bos_id
        self.pad_token_id = dims.pad_id
        self.eos_token_id = dims.eos_id
        self.lang = "<|en|>"

    def get_vocab(self):
        return dict(self.special)

    def set_prefix_tokens(self, language=None, task=None, predict_timestamps=None):
        if language:
            key = {"english": "<|en|>", "en": "<|en|>", "dutch": "<|nl|>", "nl": "<|nl|>"}.get(str(language).lower())
            self.lang = key or "<|en|>"

    @property
    def prefix_tokens(self):
        """WhisperTokenizer.prefix_tokens with predict_timestamps=False (what the reference's processors have: their
        `no_timestamps=` keyword is not a tokenizer argument): sot, language, task, <|notimestamps|>"""
        s = self.special
        return [s["<|startoftranscript|>"], s[self.lang], s["<|transcribe|>"], s["<|notimestamps|>"]]

    def encode_text(self, text: str):
        ids = [(ord(c) * 31 + 7) % self.n_text for c in text]
        s = self.special
        return [s["<|startoftranscript|>"], s[self.lang], s["<|transcribe|>"], s["<|notimestamps|>"]] + ids + [s["<|endoftext|>"]]

    def pad(self, features, return_tensors="pt"):
        L = max(len(f["input_ids"]) for f in features)
        ids = torch.full((len(features), L), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros((len(features), L), dtype=torch.long)
        for i, f in enumerate(features):
            n = len(f["input_ids"])
            ids[i, :n] = torch.tensor(f["input_ids"], dtype=torch.long)
            mask[i, :n] = 1
        return _Batch(input_ids=ids, attention_mask=mask)

    def batch_decode(self, ids, skip_special_tokens=True):
        out = []
        for row in np.asarray(ids):
            toks = [int(t) for t in row if t >= 0 and (not skip_special_tokens or int(t) < self.n_text)]
            out.append(" ".join(str(t) for t in toks))
        return out


class SyntheticProcessor:
    def __init__(self, dims: WhisperDims):
        self.tokenizer = SyntheticTokenizer(dims)
        self.feature_extractor = None

    def __call__(self, text=None, **_):
        return _Batch(input_ids=self.tokenizer.encode_text(text))

    def batch_decode(self, ids, skip_special_tokens=True):
        return self.tokenizer.batch_decode(ids, skip_special_tokens)

    def get_decoder_prompt_ids(self, language=None, task=None, no_timestamps=True):
        s = self.tokenizer.special
        return [(1, s[self.tokenizer.lang]), (2, s["<|transcribe|>"]), (3, s["<|notimestamps|>"])]


def write_synthetic_dataset(root: str, n: int, ch_file: int = 224, name: str = "gwilliams", seed: int = 0,
                            min_len: int = 700, max_len: int = 7000, fixed_chars: int | None = None):
    """n samples of shape (ch_file, len) float64 in [-1, 1] + a JSONL list in the reference's schema.
    fixed_chars: every sentence cut / padded to exactly that many characters (equal label lengths: a per-rank token mean
    then equals the global one, which the data-parallel tests rely on)."""
    os.makedirs(os.path.join(root, name), exist_ok=True)
    rng = np.random.default_rng(seed)
    rows = []
    words = ["the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "meg", "signal", "brain", "speech"]
    for i in range(n):
        L = int(rng.integers(min_len, max_len + 1))
        x = np.clip(0.35 * rng.standard_normal((ch_file, L)), -1, 1)
        path = os.path.join(root, name, f"sample_{i:05d}.npy")
        np.save(path, x)
        chosen = [str(w) for w in rng.choice(words, size=int(rng.integers(3, 9)))]
        sent = " ".join(chosen)
        if fixed_chars is not None:
            sent = (sent + " " + " ".join(words))[:fixed_chars].rstrip().ljust(fixed_chars, "x")
        # per-sentence / per-word timing records for --timestamps=True (finetune.py's default): words spread over the signal
        step = (L / 200.0) / (len(chosen) + 1)
        wrec = [{"start": round(step * (k + 0.5), 2), "end": round(step * (k + 1.4), 2), "word": w} for k, w in enumerate(chosen)]
        rows.append({"eeg": {"path": path}, "sentence": sent, "language": "English", "duration": L / 200.0,
                     "sentences": [{"start": wrec[0]["start"], "end": wrec[-1]["end"], "text": sent, "words": wrec}]})
    jl = os.path.join(root, f"{name}_data.jsonl")
    with open(jl, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    return jl
