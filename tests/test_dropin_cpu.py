"""CPU: the drop-in host layer (utils.*, finetune/evaluation flag plumbing) against golden outputs of the REFERENCE's
own reader / collator (tests/golden/reader.npz, tools/make_goldens.py reader) and against torch's samplers."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from neuspeech1_amd.synthetic import SyntheticProcessor
from neuspeech1_amd.weights import WHISPER_BASE

G = os.path.join(os.path.dirname(__file__), "golden")


def test_reader_and_collator_match_reference_outputs(tmp_path):
    from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding
    from utils.reader import CustomDataset
    g = np.load(os.path.join(G, "reader.npz"))
    proc = SyntheticProcessor(WHISPER_BASE)
    for name, ch_file, modal_ch in (("gwilliams", 224, 208), ("schoffelen", 301, 273), ("other", 100, 208)):
        os.makedirs(tmp_path / name, exist_ok=True)
        rows = []
        for n in (700, 6000, 7321):
            x = np.random.default_rng(ch_file * 100000 + n).standard_normal((ch_file, n))
            p = str(tmp_path / name / f"s{n}.npy")
            np.save(p, x)
            rows.append({"eeg": {"path": p}, "sentence": f"hello {name} {n}", "language": "English", "duration": n / 200})
        jl = str(tmp_path / f"{name}.jsonl")
        with open(jl, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
        ds = CustomDataset(data_list_path=jl, processor=proc, modal="eeg", modal_ch=modal_ch, mode="val", sample_rate=200,
                           orig_sample_rate=200, language="English", timestamps=False, min_duration=0.5, max_duration=30)
        items = [ds[i] for i in range(len(ds))]
        for i, it in enumerate(items):
            a = it["input_features"][0]
            assert a.dtype == np.float64 and tuple(g[f"{name}.{i}.shape"]) == a.shape == (modal_ch, 6000)
            assert a.sum() == float(g[f"{name}.{i}.sum"]) and np.abs(a).sum() == float(g[f"{name}.{i}.abs"])
            assert np.nonzero(np.abs(a).sum(0))[0][-1] == int(g[f"{name}.{i}.last_nz"])
            assert (np.abs(a).sum(1) > 0).sum() == int(g[f"{name}.{i}.rows_nz"])
            assert list(it["labels"]) == g[f"{name}.{i}.labels"].tolist()
        batch = DataCollatorSpeechSeq2SeqWithPadding(processor=proc)(items)
        assert str(batch["input_features"].dtype) == str(g[f"{name}.batch_dtype"]) == "torch.float32"
        assert batch["input_features"].shape == (3, modal_ch, 6000)
        assert batch["input_features"].double().sum().item() == float(g[f"{name}.batch_sum"])
        assert np.array_equal(batch["labels"].numpy(), g[f"{name}.batch_labels"])


def test_timestamp_labels_match_reference_reader(tmp_path):
    """--timestamps=True is finetune.py's DEFAULT (the training recipes of the reference do not override it): labels are
    <|sot|> <|lang|> <|task|> then <|t_start|> tokens <|t_end|> per sentence / per word on the 0.02 s grid
    (reference reader :347-400).  Ids of the reference's own reader on the same records (tests/golden/reader.npz)."""
    from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding
    from utils.reader import CustomDataset
    g = np.load(os.path.join(G, "reader.npz"))
    rows = json.loads(str(g["ts.rows"]))
    proc = SyntheticProcessor(WHISPER_BASE)
    rng = np.random.default_rng(99)
    for k, r in enumerate(rows):        # the signal side of the reader is pinned by the test above; here: the labels
        p = str(tmp_path / f"gwilliams_ts{k}.npy")
        np.save(p, rng.standard_normal((224, 1200 + 300 * k)))
        r["eeg"]["path"] = p
    jl = str(tmp_path / "ts.jsonl")
    with open(jl, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    for level in ("sentences", "words"):
        kw = dict(data_list_path=jl, processor=proc, modal="eeg", modal_ch=208, mode="val", level=level, sample_rate=200,
                  orig_sample_rate=200, language="English", timestamps=True, min_duration=0.5, max_duration=30)
        ds = CustomDataset(**kw)
        items = [ds[i] for i in range(len(ds))]
        for i, it in enumerate(items):
            assert list(it["labels"]) == g[f"ts.{level}.{i}.labels"].tolist(), (level, i)
            assert it["input_features"][0].shape == (208, 6000)
        batch = DataCollatorSpeechSeq2SeqWithPadding(processor=proc)(items)
        assert np.array_equal(batch["labels"].numpy(), g[f"ts.{level}.batch_labels"])
        # the on-GPU feed's raw mode carries the same labels
        raw = CustomDataset(raw_signals=True, **kw)
        assert [list(raw[i]["labels"]) for i in range(len(raw))] == [list(it["labels"]) for it in items]
    # odd centiseconds: starts round up, ends round down to the 0.02 s grid
    ds = CustomDataset(**kw)
    assert ds._time_token(0.13, True) == ds.timestamp_begin + 7 and ds._time_token(0.13, False) == ds.timestamp_begin + 6
    assert ds._time_token(0.12, True) == ds._time_token(0.12, False) == ds.timestamp_begin + 6


def test_reader_refuses_out_of_path_modes(tmp_path):
    from utils.reader import CustomDataset
    proc = SyntheticProcessor(WHISPER_BASE)
    jl = str(tmp_path / "x.jsonl")
    open(jl, "w").write("")
    for kw in (dict(modal="speech"), dict(combine_sentences=True), dict(split_sentences=True)):
        with pytest.raises(NotImplementedError):
            CustomDataset(data_list_path=jl, processor=proc, **{"modal": "eeg", **kw})


def test_checkpoint_rotation_keeps_the_newest_five(tmp_path):
    import finetune
    for st in (1000, 3000, 2000, 12000, 7000, 9000, 11000):
        os.makedirs(tmp_path / f"checkpoint-{st}")
    os.makedirs(tmp_path / "checkpoint-final")
    finetune.rotate_checkpoints(str(tmp_path), keep=5)      # save_total_limit=5, finetune.py:245
    assert sorted(os.listdir(tmp_path)) == sorted(["checkpoint-final"] + [f"checkpoint-{s_}" for s_ in (3000, 7000, 9000, 11000, 12000)])


def test_add_arguments_bool_and_none_parsing():
    from utils.utils import add_arguments
    p = argparse.ArgumentParser()
    add_arguments("flag", bool, False, "h", p)
    add_arguments("name", str, "x", "h", p)
    add_arguments("n", int, 3, "h", p)
    for s, v in (("True", True), ("yes", True), ("ON", True), ("1", True), ("false", False), ("No", False), ("0", False)):
        assert p.parse_args([f"--flag={s}"]).flag is v
    with pytest.raises(SystemExit):
        p.parse_args(["--flag=maybe"])
    assert p.parse_args(["--name=None"]).name is None and p.parse_args(["--name=abc"]).name == "abc"
    assert p.parse_args([]).flag is False and p.parse_args(["--n=5"]).n == 5


def test_cli_flag_names_and_defaults_match_reference():
    import evaluation
    import finetune
    ft = {a.dest: a.default for a in finetune.build_parser()._actions}
    ev = {a.dest: a.default for a in evaluation.build_parser()._actions}
    # reference finetune.py:25-64 / evaluation.py:25-49 (names and defaults; paths are site-specific)
    for k, v in dict(warmup_steps=10000, logging_steps=100, eval_steps=1000, save_steps=1000, num_workers=6,
                     learning_rate=1e-3, modal="speech", sampling_rate=200, orig_sample_rate=200, eeg_ch=224,
                     lora_eeg_ch=None, min_audio_len=0.5, max_audio_len=30, use_adalora=True, fp16=False, use_8bit=False,
                     filter_dataset=False, timestamps=True, local_files_only=True, num_train_epochs=30,
                     language="English", task="transcribe", resume_from_checkpoint=None,
                     per_device_train_batch_size=2, per_device_eval_batch_size=2, gradient_accumulation_steps=1,
                     fine_tune_layers=None, device="auto", config_name="base", data_ratio=None,
                     random_initialize_whisper=False, combine_sentences=False, split_sentences=False, ft_full=False,
                     lora_model=None, output_dir="output1/").items():
        assert ft[k] == v, k
    for k, v in dict(lora_model=None, modal="speech", sampling_rate=1000, eeg_ch=66, batch_size=16, num_workers=8,
                     language="Chinese", remove_pun=True, to_simple=True, timestamps=True, min_audio_len=0.5,
                     max_audio_len=30, local_files_only=True, noise=False, filter_dataset=False, random_choice=False,
                     task="transcribe", random_initialize_whisper=False, teacher_forcing=False, extra_name=None,
                     post_processing=False, config_name="base", add_sequence_bias=False).items():
        assert ev[k] == v, k


def test_module_names_and_lora_target_selection():
    """finetune.py:189-198: 36 adapter targets for whisper-base, selected by HF dotted names."""
    from utils.load_model import WhisperForConditionalGeneration, _cfg_from_dims, match_modules, match_modules_string
    from utils.model_utils import projection_module
    model = WhisperForConditionalGeneration(_cfg_from_dims(WHISPER_BASE))
    conv1 = projection_module(config_name="base", meg_ch=208, d_model=model.model.encoder.conv2.in_channels)
    assert conv1.stride == (2,) and list(conv1.state_dict()) == ["0.weight", "0.bias", "2.weight", "2.bias"]
    model.model.encoder.set_input_embeddings(conv1)
    t = match_modules_string(model.named_modules(), ["model.encoder"], ["k_proj", "q_proj", "v_proj", "out_proj", "fc1", "fc2"])
    assert len(t) == 36 and t[0] == "model.encoder.layers.0.self_attn.k_proj" and t[-1] == "model.encoder.layers.5.fc2"
    assert all(n.startswith("model.encoder.layers.") for n in t)
    names = [n for n, _ in model.named_parameters()]
    assert "model.encoder.conv1.0.weight" in names and "model.encoder.conv1.2.bias" in names and "model.encoder.conv2.weight" in names
    assert "model.encoder.layers.3.self_attn.k_proj.bias" not in names and "model.decoder.embed_tokens.weight" in names
    assert match_modules([("a.original_module.w", 0), ("b.w", 0)], [""], [""], ["original_module"]) == ["a.original_module.w"]
    from neuspeech1_amd.peft_compat import LoraConfig, get_peft_model
    for bad in (t[:5], t[6:12]):       # not "all six modules of the first N layers" (N = --fine_tune_layers)
        with pytest.raises(NotImplementedError):
            get_peft_model(model, LoraConfig(r=32, lora_alpha=64, target_modules=bad))
    pm = get_peft_model(model, LoraConfig(r=32, lora_alpha=64, target_modules=t, lora_dropout=0.05,
                                          modules_to_save=["model.encoder.conv1", "model.encoder.conv2"]))
    for p in pm.parameters():
        pass
    tr = sum(p.numel() for n, p in pm.named_parameters() if ".lora_" in n)
    assert tr == 1_769_472      # LoRA r32 on 36 Linear layers of whisper-base
    assert "base_model.model.model.encoder.layers.0.self_attn.q_proj.lora_A.default.weight" in dict(pm.named_parameters())


def test_product_path_never_imports_the_oracle():
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, f) for f in ("finetune.py", "evaluation.py")]
    for d in ("neuspeech1_amd", "utils"):
        files += [os.path.join(root, d, f) for f in os.listdir(os.path.join(root, d)) if f.endswith(".py")]
    for path in files:
        if path.endswith("smoke.py"):
            continue    # __graft_entry__.smoke(): the one sanctioned checker call site inside the package
        for node in ast.walk(ast.parse(open(path).read())):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or ""]
            assert not any(m == "oracle" or m.startswith("oracle.") for m in mods), path


def test_shard_indices_follow_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    from finetune import shard_indices
    data = list(range(103))
    for world in (1, 2, 8):
        for epoch in (0, 3):
            for rank in range(world):
                s = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=42)
                s.set_epoch(epoch)
                assert list(s) == shard_indices(len(data), epoch, rank, world)


def test_from_pretrained_reads_a_stock_hf_checkpoint_and_its_generation_config(tmp_path):
    """SURVEY §8 f2 (finetune.py:127-131, evaluation.py:72-74 load `openai/whisper-*`): a directory WRITTEN BY STOCK
    transformers save_pretrained (tests/hf_ckpt.py) loads tensor-for-tensor, and generation_config.json (suppress
    lists, max_length) is picked up.  No compute: the GPU side is tests/test_hf_ckpt_gpu.py."""
    from tests.hf_ckpt import GEN_CFG, write_stock_hf_checkpoint
    from neuspeech1_amd.weights import TINY
    from utils.load_model import WhisperForConditionalGeneration
    ref_sd = write_stock_hf_checkpoint(TINY, str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["config.json", "generation_config.json", "model.safetensors"]
    model = WhisperForConditionalGeneration.from_pretrained(str(tmp_path), device_map="cpu")
    own = model.state_dict()
    for k, v in ref_sd.items():
        assert k in own, k
        assert torch.equal(own[k], v), k
    assert model.model.encoder.conv1.in_channels == 80           # the stock mel front-end until the MEG one is installed
    gc = GEN_CFG(TINY)
    assert list(model.generation_config.suppress_tokens) == gc["suppress_tokens"]
    assert list(model.generation_config.begin_suppress_tokens) == gc["begin_suppress_tokens"]
    assert model.generation_config.max_length == gc["max_length"]
    assert model.config.d_model == TINY.d and model.config.encoder_attention_heads == TINY.heads
    # a save -> load round trip keeps the generation defaults
    out = tmp_path / "again"
    model.save_pretrained(str(out))
    again = WhisperForConditionalGeneration.from_pretrained(str(out), device_map="cpu")
    assert vars(again.generation_config) == vars(model.generation_config)


def test_merge_lora_copies_the_processor_files(tmp_path, monkeypatch):
    """reference merge_lora.py:24-29,50-54 saves feature extractor / tokenizer / processor into full_model so that
    evaluation.py can load WhisperProcessor from it; here their files travel from the base directory."""
    import merge_lora
    base = tmp_path / "base"
    base.mkdir()
    for name in ("tokenizer.json", "vocab.json", "preprocessor_config.json", "unrelated.bin"):
        (base / name).write_text("{}")
    assert "tokenizer.json" in merge_lora.PROCESSOR_FILES and "unrelated.bin" not in merge_lora.PROCESSOR_FILES
    import inspect
    src = inspect.getsource(merge_lora.main)
    assert "PROCESSOR_FILES" in src and "shutil.copy2" in src
