"""Batch collation for the MEG path (reference utils/data_utils.py:181-221) plus the small text helpers the CLIs
import.  The collator emits exactly what the HIP engine's ns_signal_pack consumes: float32 (B, ch, 6000)."""
import re
import string
from dataclasses import dataclass
from typing import Any, Dict, List, Union

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


_PRELOAD = ["numpy", "torch", "torch.utils.data", "transformers", "utils.reader", "utils.data_utils", "neuspeech1_amd.feed",
            "neuspeech1_amd.synthetic"]


def worker_context(num_workers: int):
    """`multiprocessing_context` for every DataLoader of the CLIs (reference worker processes: finetune.py:249,
    evaluation.py:126-127): a FORKSERVER.

    The workers must never be forked from the process that drives the GPU: a forked child inherits the parent's HIP objects,
    and ANY destructor that runs there (cyclic garbage the parent had not collected yet, an exception unwinding, interpreter
    shut-down) is a HIP call in a forked process -- segfaulting DataLoader workers in round 4, papered over with
    gc.collect() + gc.freeze() around the fork, which covers cyclic garbage only.  With a forkserver the workers are forked
    from a helper process that was started fresh (`spawn`: no HIP state, no GPU memory mappings) and has only imported the
    modules in _PRELOAD, so a worker starts in milliseconds and can never see a HIP object.  Dataset, collator and sampler
    travel by pickle.  Returns None for num_workers == 0 (DataLoader's in-process path)."""
    if num_workers <= 0:
        return None
    import multiprocessing as mp
    ctx = mp.get_context("forkserver")
    start_worker_server()
    return ctx


def start_worker_server():
    """start the forkserver (idempotent).  The CLIs call this FIRST, before the model touches the GPU: the helper is a fresh
    child process either way, but started early its imports (_PRELOAD, ~3 s) overlap the model load instead of the first batch."""
    import multiprocessing as mp
    from multiprocessing import forkserver
    import importlib.util
    mp.set_forkserver_preload([m for m in _PRELOAD if importlib.util.find_spec(m.split(".")[0]) is not None])
    forkserver.ensure_running()


class _main_not_reimported:
    """While workers start: hide `__main__`'s file / spec from multiprocessing.  A forkserver / spawn child otherwise RE-IMPORTS the
    parent's main script (as `__mp_main__`) before it unpickles its task -- harmless for `python finetune.py` (guarded by
    `if __name__ == "__main__"`), fatal for any caller that runs `finetune.main([...])` from an unguarded script or a notebook cell
    (the child re-runs the training call: "An attempt has been made to start a new process before the current process has finished its
    bootstrapping phase", seen with tools/run_recipe.py).  The workers need nothing from `__main__`: dataset and collator live in
    importable modules (utils.reader, utils.data_utils), samplers stay in the parent."""

    def __enter__(self):
        import sys
        self.m = sys.modules.get("__main__")
        self.saved = (getattr(self.m, "__file__", None), getattr(self.m, "__spec__", None), hasattr(self.m, "__file__"))
        if self.m is not None:
            self.m.__file__ = None
            self.m.__spec__ = None
        return self

    def __exit__(self, *exc):
        if self.m is not None:
            f_, spec, had = self.saved
            if had:
                self.m.__file__ = f_
            else:
                try:
                    del self.m.__file__
                except AttributeError:
                    pass
            self.m.__spec__ = spec
        return False


def fork_safe_iter(loader):
    """iter(loader).  The one place the CLIs start their workers: a loader with workers must have been built with
    `multiprocessing_context=worker_context(n)` -- forking them from this (GPU-driving) process is refused -- and the children do not
    re-import the caller's main script.  REQUIREMENT that follows: whatever the workers unpickle -- the dataset, the collate_fn, a
    worker_init_fn, a batch sampler -- must be defined in an IMPORTABLE module (utils.reader / utils.data_utils are), not in the caller's
    main script or notebook cell: a class whose __module__ is '__main__' cannot be found by a child that does not re-import the script.
    Checked up front, so the failure is this message and not a pickling traceback inside a worker."""
    if getattr(loader, "num_workers", 0) > 0:
        ctx = getattr(loader, "multiprocessing_context", None)
        method = ctx.get_start_method() if ctx is not None else None
        if method not in ("forkserver", "spawn"):
            raise RuntimeError("DataLoader workers must come from utils.data_utils.worker_context() (forkserver), never from a fork "
                               f"of the process that holds the GPU (start method: {method or 'fork (default)'})")
        local = []
        for what in ("dataset", "collate_fn", "worker_init_fn", "batch_sampler", "sampler"):
            obj = getattr(loader, what, None)
            mod = getattr(obj, "__module__", None)      # functions and classes carry their own; an instance answers with its class's
            if obj is not None and mod == "__main__":
                local.append(f"{what} ({getattr(obj, '__qualname__', type(obj).__qualname__)})")
        if local:
            raise RuntimeError("DataLoader workers are started without re-importing the caller's main script, so they cannot unpickle objects "
                               "defined in it: " + ", ".join(local) + ".  Move them into an importable module (as utils.reader.CustomDataset "
                               "and utils.data_utils.DataCollatorBrainSpeechSeq2SeqWithPadding are), or use num_workers=0.")
        with _main_not_reimported():
            return iter(loader)
    return iter(loader)
