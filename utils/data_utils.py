"""Batch collation for the MEG path (reference utils/data_utils.py:181-221) plus the small text helpers the CLIs
import.  The collator emits exactly what the HIP engine's ns_signal_pack consumes: float32 (B, ch, 6000)."""
import re
import string
from dataclasses import dataclass
from typing import Any, Dict, List, Union

import gc

import numpy as np
import torch


def get_part_of_dataset(dataset, ratio):
    return dataset[: int(ratio * len(dataset))]


def generate_random_string(length):
    chars = list(string.ascii_letters + string.digits)
    return "".join(np.random.choice(chars, length))


def remove_punctuation(text):
    punctuation = "!,.;:?、！，。；：？"
    if isinstance(text, str):
        return re.sub(r"[{}]+".format(punctuation), "", text).strip()
    if isinstance(text, list):
        return [re.sub(r"[{}]+".format(punctuation), "", t).strip() for t in text]
    raise Exception(f"unsupported type {type(text)}")


def to_simple(text):
    try:
        from zhconv import convert
    except ImportError as e:  # zhconv is not part of the image; only the Chinese post-processing needs it
        raise ImportError("to_simple() needs the `zhconv` package") from e
    if isinstance(text, str):
        return convert(text, "zh-cn")
    return [convert(t, "zh-cn") for t in text]


def contains_valid_letters(s, prefix="Ġ", biaodian=",.'`:?"):
    if len(s) < 1:
        return False
    if prefix == s[0]:
        s = s[1:]
    return re.match(f"^[A-Za-z{biaodian}]+$", s) is not None


@dataclass
class DataCollatorSpeechSeq2SeqWithPadding:
    """features: [{'input_features': [np (ch, 6000)], 'labels': [ids]}] ->
    {'input_features': float32 (B, ch, 6000), 'labels': int64 (B, L) with pad -> -100}.
    A leading BOS column is stripped only when EVERY row starts with tokenizer.bos_token_id (reference :217-218)."""
    processor: Any
    vocab_size: int = 51865

    def __call__(self, features: List[Dict[str, Union[List[int], torch.Tensor]]]) -> Dict[str, torch.Tensor]:
        # one pinned float32 block; each sample is converted (f64 -> f32) straight into its slot
        if type(features[0]["input_features"][0]).__name__ == "RawSignal":
            # on-GPU feed (CustomDataset(raw_signals=True)): the recordings are described, not loaded
            batch = {"input_features": [f["input_features"][0] for f in features]}
        else:
            first = np.asarray(features[0]["input_features"][0])
            x = torch.empty((len(features),) + first.shape, dtype=torch.float32)
            for i, f in enumerate(features):
                x[i] = torch.from_numpy(np.ascontiguousarray(f["input_features"][0]))
            batch = {"input_features": x}
        tok = self.processor.tokenizer
        label_features = [{"input_ids": f["labels"]} for f in features]
        labels_batch = tok.pad(label_features, return_tensors="pt")
        labels = labels_batch["input_ids"].masked_fill(labels_batch["attention_mask"].ne(1), -100)
        if int(labels.max()) >= self.vocab_size:
            bad = labels[labels >= self.vocab_size]
            print(f"input_ids beyond the vocabulary ({self.vocab_size}): {bad.tolist()[:16]} ({bad.numel()} total)")
        if bool((labels[:, 0] == tok.bos_token_id).all()):
            labels = labels[:, 1:]
        batch["labels"] = labels
        return batch


# the reference's speech-only collator is the same object (reference :150-178)
DataCollatorOnlySpeechSeq2SeqWithPadding = DataCollatorSpeechSeq2SeqWithPadding


def fork_safe_iter(loader):
    """iter(loader) for a DataLoader whose workers are FORKED from a process that holds HIP objects.

    A forked child must not run a HIP call.  The workers' own code does not -- but their cyclic garbage collector may: garbage the parent
    has not collected yet (a closed feed's staging slots with their events and streams, a finished generator's graph objects) is garbage
    in the child too, and the first collection there runs the destructors, i.e. hipEventDestroy / hipStreamDestroy in a forked process:
    a segmentation fault in a DataLoader worker, depending on allocation counts (seen in round 4 when a shape change moved the collector's
    threshold).  Collect in the parent first, then freeze what is left for the duration of the fork (gc.freeze() exists for this: the
    children inherit the frozen generation and never look at it)."""
    if getattr(loader, "num_workers", 0) > 0:
        gc.collect()
        gc.freeze()
        try:
            return iter(loader)      # _MultiProcessingDataLoaderIter starts (forks) its workers here
        finally:
            gc.unfreeze()
    return iter(loader)

