"""JSONL -> (ch, 6000) MEG array -> labels (reference utils/reader.py, EEG branch :128-524).

Kept: the constructor signature, `.data_list`, `__len__`, `__getitem__`, the dataset-specific channel slice
(schoffelen [28:301], gwilliams [:208]), channel zero-padding, time crop / right zero-padding to
max_duration*sample_rate, timestamp-free labels = processor(text=...).input_ids, timestamp labels (finetune.py's default
--timestamps=True: <|t_start|> text <|t_end|> per sentence or word on the 0.02 s grid) and the <|nocaptions|> fallback.
Out of the hot path and therefore refused loudly: modal='speech', combine_sentences, split_sentences and the noise /
shift augmentations (their shipped probabilities are 0.0, configs/augmentation1.json).
"""
import copy
import json
from typing import List

import numpy as np
from torch.utils.data import Dataset


def read_jsonlines(path):
    with open(path, "r", encoding="utf-8") as f:
        return [json.loads(line) for line in f if line.strip()]


def write_jsonlines(path, rows):
    with open(path, "w", encoding="utf-8") as f:
        for r in rows:
            f.write(json.dumps(r, ensure_ascii=False) + "\n")


class CustomDataset(Dataset):
    def __init__(self, data_list_path, processor, data_list_dir="", mode="train", modal="eeg", modal_ch=66,
                 level="sentences", language=None, filter_dataset=False, timestamps=False, sample_rate=200,
                 orig_sample_rate=200, min_duration=0.5, max_duration=30, combine_sentences=False,
                 split_sentences=False, subj=None, augment_config_path=None, raw_signals=False):
        assert min_duration >= 0.5, f"min_duration must be >= 0.5, got {min_duration}"
        assert max_duration <= 30, f"max_duration must be <= 30, got {max_duration}"
        if modal != "eeg":
            raise NotImplementedError("only modal='eeg' (MEG/EEG arrays) is on the MI355X hot path")
        if combine_sentences or split_sentences:
            raise NotImplementedError("combine_sentences / split_sentences are outside the hot path")
        self.data_list_path, self.processor, self.mode, self.level = data_list_path, processor, mode, level
        self.signal_sample_rate, self.orig_sample_rate = sample_rate, orig_sample_rate
        self.language, self.filter_dataset, self.timestamps = language, filter_dataset, timestamps
        self.data_list_dir, self.modal, self.modal_ch = data_list_dir, modal, modal_ch
        self.min_duration, self.max_duration, self.subj = min_duration, max_duration, subj
        # raw_signals (addition of this build): __getitem__ returns a neuspeech1_amd.feed.RawSignal (file + channel
        # slice) instead of the loaded array; the slice / pad / crop / cast rules below then run on the GPU (ns_feed_pack)
        self.raw_signals = raw_signals
        vocab = processor.tokenizer.get_vocab()
        self.startoftranscript = vocab["<|startoftranscript|>"]
        self.endoftext = vocab["<|endoftext|>"]
        self.nocaptions = vocab["<|nocaptions|>"]
        self.timestamp_begin = vocab["<|notimestamps|>"] + 1
        self.data_list: List[dict] = []
        self._load_data_list()
        self.augment_configs = None
        if augment_config_path:
            with open(augment_config_path, "r", encoding="utf-8") as f:
                self.augment_configs = json.load(f)
            for k, v in self.augment_configs.items():
                if isinstance(v, dict) and float(v.get("prob", 0.0)) > 0.0:
                    raise NotImplementedError(f"augmentation '{k}' with prob > 0 is outside the hot path")

    def _load_data_list(self):
        rows = read_jsonlines(self.data_list_path)
        if self.filter_dataset:
            rows = [r for r in rows if r.get("sent_type") == "ZINNEN" and r["duration"] < 30]
        if self.subj is not None:
            rows = [r for r in rows if r.get("subj") == self.subj]
        self.data_list = rows
        print(f"num of data:{len(self.data_list)} mode:{self.mode}")

    def _get_list_data(self, idx):
        row = copy.deepcopy(self.data_list[idx])
        path = row[self.modal]["path"]
        assert path is not None
        lo, hi = self.channel_slice(path)
        # timestamp labels are built from the per-sentence records (reference :265); finetune.py defaults to them
        transcript = row["sentences"] if self.timestamps else row["sentence"]
        if self.raw_signals:
            from neuspeech1_amd.feed import RawSignal
            return RawSignal(path, lo, hi, self.modal_ch), self.signal_sample_rate, transcript, row.get("language")
        sample = np.load(path)[lo:hi]                # (>=ch, n) float64
        if self.modal_ch > sample.shape[0]:
            sample = self.pad_sample_ch(sample)
        return sample, self.signal_sample_rate, transcript, row.get("language")

    def channel_slice(self, path):
        """dataset-specific channel rows (reference :282-290)"""
        if "schoffelen" in path:
            return 28, 301
        if "gwilliams" in path:
            return 0, 208
        return 0, self.modal_ch

    def __getitem__(self, idx):
        sample, _, transcript, language = self._get_list_data(idx)
        self.processor.tokenizer.set_prefix_tokens(language=language if language is not None else self.language)
        if len(transcript) > 0:
            labels = self._load_timestamps_transcript(transcript) if self.timestamps else self.process_transcript(transcript)
            return {"input_features": self.padding_sample(sample), "labels": labels}
        return {"input_features": self.padding_sample(sample),
                "labels": [self.startoftranscript, self.nocaptions, self.endoftext]}

    def padding_sample(self, sample):
        """crop to max_duration*sample_rate samples, zero-pad on the right (reference :496-506)."""
        if self.raw_signals:
            return [sample]
        max_length = int(self.max_duration * self.signal_sample_rate)
        sample = sample[:, :max_length]
        sample = np.pad(sample, pad_width=((0, 0), (0, max_length - sample.shape[-1])))
        assert sample.shape == (self.modal_ch, max_length), f"sample shape {sample.shape} != {(self.modal_ch, max_length)}"
        return [sample]

    def pad_sample_ch(self, sample):
        """zero channels appended up to modal_ch (reference :508-516)."""
        assert sample.ndim == 2, f"sample.shape is {sample.shape}"
        if sample.shape[0] == self.modal_ch:
            return sample
        assert sample.shape[0] < self.modal_ch, "sample channel must be less than modal channel"
        return np.pad(sample, pad_width=((0, self.modal_ch - sample.shape[0]), (0, 0)))

    # ---- timestamp labels (reference :347-400): <|sot|> <|lang|> <|task|> then per segment
    #      <|t_start|> text tokens <|t_end|>, times on Whisper's 0.02 s grid, <|endoftext|> last
    def _time_token(self, t, is_start):
        if round(t * 100) % 2 != 0:
            t = t + 0.01 if is_start else t - 0.01
        return self.timestamp_begin + round(t * 100) // 2

    def _segment(self, labels, start, end, text):
        s_tok, e_tok = self._time_token(start, True), self._time_token(end, False)
        label = self.processor(text=text).input_ids[4:-1]
        if max(label) > 51865 or s_tok > 51865 or e_tok > 51865:
            print(f"OOV text {text} label {label} start {start}->{s_tok} end {end}->{e_tok}\n")
            raise ValueError
        labels.extend([s_tok])
        labels.extend(label)
        labels.extend([e_tok])

    def _load_timestamps_transcript(self, transcript):
        assert isinstance(transcript, list), f"transcript must be a list, got {type(transcript)}"
        labels = list(self.processor.tokenizer.prefix_tokens[:3])
        if self.level == "sentences":
            for t in transcript:
                self._segment(labels, t["start"], t["end"], t["text"])
        elif self.level == "words":
            for t in transcript:
                for w in t["words"]:
                    self._segment(labels, w["start"], w["end"], w["word"])
        else:
            raise NotImplementedError
        return labels + [self.endoftext]

    def process_transcript(self, transcript):
        return self.processor(text=transcript)["input_ids"]

    def __len__(self):
        return len(self.data_list)
