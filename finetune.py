"""Fine-tune Whisper on MEG with LoRA + conv-stem training on MI355X (drop-in for the reference finetune.py:
same flags, same defaults, same outputs), driven by the HIP engine instead of HF Trainer + peft:

    python finetune.py --base_model=synthetic:base --modal=eeg --eeg_ch=208 --use_adalora=False --fp16=True ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 finetune.py ...   (data parallel)

Per step (reference finetune.py:231-281 -> HF Trainer inner loop): forward, backward, RCCL all-reduce (AVG) of the
flat trainable-gradient buffer overlapped with backward, GradScaler unscale + inf check, clip to 1.0, AdamW, linear
warmup/decay; eval = loss only; a checkpoint is written at a save step only while the latest recorded eval loss is the best so far
(utils/callback.py:11-32) plus `checkpoint-final` at the end.
"""
import argparse
import os as _os
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL across processes on this pool
import functools
import json
import math
import os
import time

import torch

from neuspeech1_amd.dp import GradReducer
from neuspeech1_amd.engine import TrainCfg
from neuspeech1_amd.peft_compat import AdaLoraConfig, LoraConfig, PeftModel, get_peft_model, prepare_model_for_kbit_training
from utils.data_utils import (DataCollatorSpeechSeq2SeqWithPadding, fork_safe_iter, get_part_of_dataset, start_worker_server,
                              worker_context)
from utils.load_model import WhisperForConditionalGeneration, match_modules, match_modules_string
from utils.model_utils import projection_module
from utils.reader import CustomDataset
from utils.utils import add_arguments, make_inputs_require_grad, print_arguments


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    add_arg = functools.partial(add_arguments, argparser=parser)
    add_arg("train_data", type=str, default="dataset/train_data.jsonl", help="training list (jsonl)")
    add_arg("test_data", type=str, default="dataset/val_data.jsonl", help="validation list (jsonl)")
    add_arg("base_model", type=str, default="openai/whisper-base", help="Whisper base model dir or synthetic:<size>")
    add_arg("lora_model", type=str, default=None, help="trained adapter to merge before training")
    add_arg("output_dir", type=str, default="output1/", help="checkpoint directory")
    add_arg("warmup_steps", type=int, default=10000, help="warm-up steps")
    add_arg("logging_steps", type=int, default=100, help="log every N steps")
    add_arg("eval_steps", type=int, default=1000, help="evaluate every N steps")
    add_arg("save_steps", type=int, default=1000, help="checkpoint every N steps (only when eval loss is best)")
    add_arg("num_workers", type=int, default=6, help="data loader workers")
    add_arg("learning_rate", type=float, default=1e-3, help="learning rate")
    add_arg("modal", type=str, default="speech", help="input modality")
    add_arg("sampling_rate", type=int, default=200, help="expected signal sample rate")
    add_arg("orig_sample_rate", type=int, default=200, help="signal sample rate on disk")
    add_arg("eeg_ch", type=int, default=224, help="input channels")
    add_arg("lora_eeg_ch", type=int, default=None, help="input channels of the adapter being merged")
    add_arg("min_audio_len", type=float, default=0.5, help="minimum length (s)")
    add_arg("max_audio_len", type=float, default=30, help="maximum length (s)")
    add_arg("use_adalora", type=bool, default=True, help="AdaLoRA instead of LoRA")
    add_arg("fp16", type=bool, default=False, help="fp16 training with dynamic loss scaling")
    add_arg("use_8bit", type=bool, default=False, help="8-bit base model")
    add_arg("filter_dataset", type=bool, default=False, help="filter the data list")
    add_arg("timestamps", type=bool, default=True, help="use timestamp labels")
    add_arg("local_files_only", type=bool, default=True, help="never download")
    add_arg("num_train_epochs", type=int, default=30, help="epochs")
    add_arg("language", type=str, default="English", help="language (None = multilingual)")
    add_arg("task", type=str, default="transcribe", choices=["transcribe", "translate"], help="task")
    add_arg("augment_config_path", type=str, default="configs/augmentation.json", help="augmentation config")
    add_arg("resume_from_checkpoint", type=str, default=None, help="adapter checkpoint to resume from")
    add_arg("per_device_train_batch_size", type=int, default=2, help="train batch per device")
    add_arg("per_device_eval_batch_size", type=int, default=2, help="eval batch per device")
    add_arg("gradient_accumulation_steps", type=int, default=1, help="gradient accumulation")
    add_arg("fine_tune_layers", type=int, default=None, help="adapt only the first N encoder layers")
    add_arg("device", type=str, default="auto", help="device")
    add_arg("config_name", type=str, default="base", help="conv1 module")
    add_arg("data_ratio", type=float, default=None, help="fraction of the training list to use")
    add_arg("random_initialize_whisper", type=bool, default=False, help="random init")
    add_arg("combine_sentences", type=bool, default=False, help="sentence combining")
    add_arg("split_sentences", type=bool, default=False, help="sentence splitting")
    add_arg("ft_full", type=bool, default=False, help="adapt the whole model")
    # additions of this build (not in the reference)
    add_arg("lora_dropout", type=float, default=None, help="adapter dropout (default: the reference's 0.05 LoRA / 0.1 AdaLoRA)")
    add_arg("max_steps", type=int, default=-1, help="stop after N optimizer steps (smoke runs)")
    add_arg("device_feed", type=bool, default=True, help="slice / pad / cast the recordings on the GPU (ns_feed_pack) "
                                                        "instead of in the data-loader workers")
    add_arg("feed_cache_dir", type=str, default="", help="with --device_feed: keep the kept channel rows of every recording in this directory, "
            "rounded once to --feed_cache_dtype (bit-identical batches, a quarter / half of the float64 bytes read and staged per step)")
    add_arg("feed_cache_dtype", type=str, default="f16", help="f16 | f32 (see --feed_cache_dir)")
    return parser


def get_processor(name, language, task, timestamps, local_files_only):
    if name.startswith("synthetic:"):
        from neuspeech1_amd.synthetic import SyntheticProcessor
        from utils.load_model import _SYNTH
        p = SyntheticProcessor(_SYNTH[name.split(":")[1]])
        p.tokenizer.set_prefix_tokens(language=language)
        return p
    cfg_path = os.path.join(name, "config.json")
    if os.path.exists(cfg_path):      # a model exported from a synthetic base (merge_lora.py / save_pretrained)
        syn = json.load(open(cfg_path)).get("synthetic_name")
        if syn:
            return get_processor(f"synthetic:{syn}", language, task, timestamps, local_files_only)
    from transformers import WhisperProcessor
    return WhisperProcessor.from_pretrained(name, language=language, task=task, no_timestamps=not timestamps,
                                            local_files_only=local_files_only)


def shard_indices(n, epoch, rank, world, seed=42, shuffle=True):
    """torch DistributedSampler semantics: seeded shuffle, pad to a multiple of world, strided shard."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist() if shuffle else list(range(n))
    total = math.ceil(n / world) * world
    idx += idx[: total - n]
    return idx[rank:total:world]


class EpochShardSampler(torch.utils.data.Sampler):
    """This rank's shard of epoch `epoch` (set_epoch before every pass).  One DataLoader with persistent workers then
    serves the whole run: rebuilding the loader per epoch cost ~3.7 s of worker start-up per epoch on the GPU box."""

    def __init__(self, n, rank, world):
        self.n, self.rank, self.world, self.epoch = n, rank, world, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        return iter(shard_indices(self.n, self.epoch, self.rank, self.world))

    def __len__(self):
        return math.ceil(self.n / self.world)


def evaluate_loss(model, dataset, collator, batch_size, rank, world, num_workers, feed=None):
    """eval_loss as HF Trainer reports it: the batch losses weighted by their sample counts (evaluation_loop repeats
    each batch loss batch_size times before averaging), ranks sharded like DistributedSampler without shuffling.
    With `feed` (SignalFeed) the recordings take the on-GPU path, the next batch staged while this one runs."""
    eng = model.engine()
    idx = shard_indices(len(dataset), 0, rank, world, shuffle=False)
    chunks = [idx[i:i + batch_size] for i in range(0, len(idx), batch_size)]
    was_raw = dataset.raw_signals
    dataset.raw_signals = feed is not None
    try:
        def produce(ch):
            batch = collator([dataset[j] for j in ch])
            x = batch["input_features"]
            return (feed.submit(x) if feed is not None else x.to(model.device)), batch["labels"].to(model.device)
        tot, cnt = 0.0, 0
        nxt = produce(chunks[0]) if chunks else None
        for k, ch in enumerate(chunks):
            (x, y), nxt = nxt, (produce(chunks[k + 1]) if k + 1 < len(chunks) else None)
            if feed is not None:
                x = x.result()
            loss, _ = eng.forward(x, y, train=False)
            if feed is not None:
                x.release()
            tot += loss.item() * len(ch)
            cnt += len(ch)
    finally:
        dataset.raw_signals = was_raw
    t = torch.tensor([tot, cnt], device=model.device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t)
    return (t[0] / t[1].clamp_min(1)).item()


class DevicePrefetcher:
    """Host -> device feed of the collated batches on a copy stream, one batch ahead of the step that consumes it.
    Yields (input_features, labels) on the device.  Two sources:
      * float32 (B, ch, T) tensors from the collator (pinned, so the 319 MB batch of BASELINE configs[1] moves under
        the previous step instead of in front of this one);
      * lists of RawSignal (CustomDataset(raw_signals=True)): the recordings' bytes go through
        neuspeech1_amd.feed.SignalFeed and arrive as a PackedSignal -- release() it once its step is enqueued."""

    def __init__(self, loader, device, feed=None):
        self.loader, self.device, self.feed = loader, device, feed
        self.stream = torch.cuda.Stream(device) if device.type == "cuda" else None

    def _load(self, it):
        batch = next(it, None)
        if batch is None:
            return None
        x = batch["input_features"]
        if isinstance(x, list):
            assert self.feed is not None, "RawSignal batches need a SignalFeed"
            with torch.cuda.stream(self.stream):
                y = batch["labels"].pin_memory().to(self.device, non_blocking=True)
            return self.feed.submit(x), y       # Future: the reads run on the feed's loader thread
        if self.stream is None:
            return x.to(self.device), batch["labels"].to(self.device)
        with torch.cuda.stream(self.stream):
            return x.to(self.device, non_blocking=True), batch["labels"].to(self.device, non_blocking=True)

    def __iter__(self):
        it = fork_safe_iter(self.loader)
        nxt = self._load(it)
        while nxt is not None:
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
                if not isinstance(nxt[0], torch.Tensor):
                    nxt = (nxt[0].result().acquire(), nxt[1])
                for t in nxt:
                    if isinstance(t, torch.Tensor):
                        t.record_stream(torch.cuda.current_stream())
            cur, nxt = nxt, self._load(it)
            yield cur


def rotate_checkpoints(output_dir, keep):
    """HF Trainer's save_total_limit: only the `keep` newest checkpoint-<step> directories stay"""
    import re
    import shutil
    steps = sorted(int(m.group(1)) for m in (re.fullmatch(r"checkpoint-(\d+)", n) for n in os.listdir(output_dir)) if m)
    for st in steps[:-keep] if keep > 0 else []:
        shutil.rmtree(os.path.join(output_dir, f"checkpoint-{st}"), ignore_errors=True)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.num_workers > 0:
        start_worker_server()       # the DataLoader workers' forkserver: a fresh helper process, started before anything touches the GPU
    print_arguments(args)
    if args.gradient_accumulation_steps < 1:
        raise ValueError("gradient_accumulation_steps must be >= 1")
    processor = get_processor(args.base_model, args.language, args.task, args.timestamps, args.local_files_only)
    ds_kw = dict(processor=processor, modal=args.modal, modal_ch=args.eeg_ch, sample_rate=args.sampling_rate,
                 orig_sample_rate=args.orig_sample_rate, language=args.language, filter_dataset=args.filter_dataset,
                 timestamps=args.timestamps, min_duration=args.min_audio_len, max_duration=args.max_audio_len)
    train_dataset = CustomDataset(data_list_path=args.train_data, mode="train", combine_sentences=args.combine_sentences,
                                  split_sentences=args.split_sentences, augment_config_path=args.augment_config_path,
                                  **ds_kw)
    test_dataset = CustomDataset(data_list_path=args.test_data, mode="val", **ds_kw)
    train_dataset.raw_signals = bool(args.device_feed) and torch.cuda.is_available() and args.device != "cpu"
    if args.data_ratio is not None:
        train_dataset.data_list = get_part_of_dataset(train_dataset.data_list, args.data_ratio)
    print(f"train samples: {len(train_dataset)}, eval samples: {len(test_dataset)}")
    data_collator = DataCollatorSpeechSeq2SeqWithPadding(processor=processor)

    # one process per GPU (reference :115-122: WORLD_SIZE / LOCAL_RANK)
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK") or 0)
    ddp = world != 1 and args.device != "cpu"
    device_map = {"": local} if ddp else args.device
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL in production; NS_DIST_BACKEND=gloo lets two ranks share ONE GPU in the tests (RCCL refuses that)
        backend = os.environ.get("NS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
    print(f"device map :{device_map}")
    model = WhisperForConditionalGeneration.from_pretrained(args.base_model, load_in_8bit=args.use_8bit,
                                                            device_map=device_map, local_files_only=args.local_files_only)
    print(f"model device {model.device}")
    eeg_ch = args.lora_eeg_ch if args.lora_eeg_ch is not None else args.eeg_ch
    d_model = model.model.encoder.conv2.in_channels
    torch.manual_seed(42)
    conv1 = projection_module(config_name=args.config_name, meg_ch=eeg_ch, d_model=d_model).to(model.device)
    model.model.encoder.set_input_embeddings(conv1)
    if args.lora_model is not None:
        model = PeftModel.from_pretrained(model, args.lora_model, local_files_only=args.local_files_only).merge_and_unload()
        if args.lora_eeg_ch != args.eeg_ch:
            conv1 = projection_module(config_name=args.config_name, meg_ch=args.eeg_ch, d_model=d_model).to(model.device)
            model.model.encoder.set_input_embeddings(conv1)
    if args.random_initialize_whisper:
        model.post_init()       # reference :166-168
        print("model re-initialised at random")
    model.config.forced_decoder_ids = None
    model.config.suppress_tokens = []
    model = prepare_model_for_kbit_training(model)
    model.model.encoder.conv1.register_forward_hook(make_inputs_require_grad)
    for p in model.parameters():
        p.requires_grad = False

    if args.resume_from_checkpoint:
        print("Loading adapters from checkpoint.")
        model = PeftModel.from_pretrained(model, args.resume_from_checkpoint, is_trainable=True)
        for n in ("model.encoder.conv1", "model.encoder.conv2"):
            for p in model.model.get_submodule(n).parameters():
                p.requires_grad = True
    else:
        if args.fine_tune_layers is not None:
            prefixes = [f"model.encoder.layers.{i}." for i in range(args.fine_tune_layers)]
        elif args.ft_full:
            prefixes = ["model"]
        else:
            prefixes = ["model.encoder"]
        suffixes = ["k_proj", "q_proj", "v_proj", "out_proj", "fc1", "fc2"]
        target_modules = match_modules_string(model.named_modules(), prefixes, suffixes)
        modules_to_save = ["model.encoder.conv1", "model.encoder.conv2"]
        print("target_modules", target_modules)
        print("modules_to_save", modules_to_save)
        if args.use_adalora:
            config = AdaLoraConfig(init_r=12, target_r=4, beta1=0.85, beta2=0.85, tinit=200, tfinal=1000, deltaT=10,
                                   lora_alpha=32, lora_dropout=0.1 if args.lora_dropout is None else args.lora_dropout, orth_reg_weight=0.5, target_modules=target_modules,
                                   modules_to_save=modules_to_save)
        else:
            config = LoraConfig(r=32, lora_alpha=64, target_modules=target_modules,
                                lora_dropout=0.05 if args.lora_dropout is None else args.lora_dropout, bias="none",
                                modules_to_save=modules_to_save)
        model = get_peft_model(model, config)
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print(f"Total trainable parameters: {trainable}")
    frozen = set(match_modules(model.named_parameters(), [""], [""], ["original_module"]))
    for name, p in model.named_parameters():
        if name in frozen:
            p.requires_grad = False
    model.print_trainable_parameters()

    base = args.base_model[:-1] if args.base_model.endswith("/") else args.base_model
    output_dir = os.path.join(args.output_dir, os.path.basename(base).replace(":", "_"))
    os.makedirs(output_dir, exist_ok=True)
    B = args.per_device_train_batch_size
    accum = args.gradient_accumulation_steps
    steps_per_epoch = math.ceil(math.ceil(math.ceil(len(train_dataset) / world) / B) / accum)
    total_steps = steps_per_epoch * args.num_train_epochs
    if args.max_steps > 0:
        total_steps = min(total_steps, args.max_steps)
    whisper = model.model
    if not args.fp16 and rank == 0:
        print("note: the HIP path always computes the GEMM / attention operands in fp16 with fp32 accumulation and an fp32 "
              "residual stream (what --fp16=True gives the reference); --fp16=False only switches the dynamic loss scaler "
              "off.  Every recipe of the reference trains with --fp16=True.")
    whisper.train_cfg = TrainCfg(lr=args.learning_rate, warmup_steps=args.warmup_steps, total_steps=total_steps,
                                 fp16_scaler=args.fp16)
    whisper.config.use_cache = False
    eng = whisper.engine()
    eng.drop_seed = 42 + rank
    reducer = GradReducer(eng.G) if ddp else None

    feed = None
    if train_dataset.raw_signals:
        from neuspeech1_amd.feed import SignalFeed
        feed = SignalFeed(whisper.device, eng.dims.ch, eng.dims.T, eng.dims.ch_pad, threads=max(2, args.num_workers),
                          cache_dir=args.feed_cache_dir or None, cache_dtype=args.feed_cache_dtype)

    step, eval_history, t_log, n_log = 0, [], time.time(), 0
    log_path = os.path.join(output_dir, "train_log.jsonl")
    done = False
    loss_sum, loss_cnt = torch.zeros((), device=whisper.device), 0
    sampler = EpochShardSampler(len(train_dataset), rank, world)
    loader = torch.utils.data.DataLoader(train_dataset, batch_size=B, sampler=sampler, num_workers=args.num_workers,
                                         collate_fn=data_collator, drop_last=False, pin_memory=True,
                                         persistent_workers=args.num_workers > 0,
                                         multiprocessing_context=worker_context(args.num_workers))
    n_batches = len(loader)
    rk = dict(on_ready=reducer.on_ready, reduce_fn=reducer.finish) if reducer is not None else {}
    for epoch in range(args.num_train_epochs):
        sampler.set_epoch(epoch)
        mi, micro = 0, []
        for bi, (x, y) in enumerate(DevicePrefetcher(loader, whisper.device, feed)):
            # micro-batches are consumed as they arrive (never more than the prefetched one held beside the running one,
            # whatever --gradient_accumulation_steps is); the last group of an epoch may be shorter, as in HF Trainer
            count = min(accum, n_batches - (bi - mi))
            if count == 1:
                loss = eng.train_step(x, y, **rk)
            else:       # gradient accumulation: the exchange and the optimizer run with the last micro-batch
                micro.append(eng.accumulate_step(x, y, mi, count, **rk))
            n_log += x.shape[0]
            if hasattr(x, "release"):
                x.release()             # the step reading this staged batch is enqueued: its slot may be refilled
            mi += 1
            if mi < count:
                continue
            if count > 1:
                loss = torch.stack(micro).mean()
            mi, micro = 0, []
            step += 1
            loss_sum, loss_cnt = loss_sum + loss.detach().reshape(()).float(), loss_cnt + 1   # stays on the device
            if step % args.logging_steps == 0 and ddp:
                # HF Trainer logs the loss averaged over the ranks (`_nested_gather(tr_loss).mean()`), not rank 0's shard
                if torch.distributed.get_backend() == "gloo":
                    t_ = loss_sum.cpu()
                    torch.distributed.all_reduce(t_)
                    loss_sum = t_.to(whisper.device) / world
                else:
                    torch.distributed.all_reduce(loss_sum)
                    loss_sum = loss_sum / world
            if step % args.logging_steps == 0 and rank == 0:
                # HF Trainer logs the MEAN training loss of the steps since the previous log line
                mean_loss = (loss_sum / loss_cnt).item()        # waits for the interval's kernels: dt below is a GPU rate
                dt = time.time() - t_log
                rec = {"step": step, "epoch": round(step / steps_per_epoch, 3), "loss": round(mean_loss, 5),
                       "samples_per_s": round(world * n_log / dt, 2), "loss_scale": eng.loss_scale_dev.item()}
                print(json.dumps(rec), flush=True)
                with open(log_path, "a") as f:
                    f.write(json.dumps(rec) + "\n")
                t_log, n_log = time.time(), 0
            if step % args.logging_steps == 0:
                loss_sum, loss_cnt = torch.zeros((), device=whisper.device), 0
            # SavePeftModelCallback.on_step_end (utils/callback.py:12-22) decides BEFORE this step's evaluation: save at a
            # multiple of save_steps iff the most recent eval loss on record is the minimum of all recorded ones
            should_save = step % args.save_steps == 0 and len(eval_history) > 0 and eval_history[-1] == min(eval_history)
            if step % args.eval_steps == 0:
                ev = evaluate_loss(whisper, test_dataset, data_collator, args.per_device_eval_batch_size, rank, world,
                                   args.num_workers, feed=feed)
                eval_history.append(ev)
                if rank == 0:
                    print(json.dumps({"step": step, "eval_loss": round(ev, 5)}), flush=True)
            if should_save and rank == 0:
                model.save_pretrained(os.path.join(output_dir, f"checkpoint-{step}"))
                rotate_checkpoints(output_dir, keep=5)          # save_total_limit=5 (finetune.py:245)
            if step >= total_steps:
                done = True
                break
        if done:
            break
    if feed is not None:
        feed.close()        # its loader / reader threads end here, not whenever the collector finds the feed
    if rank == 0:
        model.save_pretrained(os.path.join(output_dir, "checkpoint-final"))
        print(f"saved {os.path.join(output_dir, 'checkpoint-final')}")
    if ddp:
        # replicas must have stayed identical: one line per rank for the logs (and for tests/test_cli_gpu.py)
        print(f"[rank {rank}] trainable checksum {eng.P.double().sum().item():.9e} after {step} steps", flush=True)
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
