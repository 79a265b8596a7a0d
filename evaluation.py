"""Decode a MEG test list with beam search on MI355X (drop-in for the reference evaluation.py: same flags, same
result files `<lora_model>/formal_test_results*.{txt,jsonl,json}`).

The decode call is the reference's (evaluation.py:369-386): do_sample=False, num_beams=5, repetition_penalty=5.0,
no_repeat_ngram_size=2, decoder_input_ids = labels[:, :4] for non-English data; --teacher_forcing takes the
argmax of one teacher-forced forward (evaluation.py:392-403).  Text metrics (BLEU/ROUGE/WER/...: `evaluate`,
`torchmetrics`, `nltk`, ... are not in the image) are out of the hot path: the json result holds throughput and
token-level accuracy instead.
"""
import argparse
import functools
import json
import os
import time

import numpy as np
import torch

from neuspeech1_amd.peft_compat import PeftModel
from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding, fork_safe_iter, start_worker_server, worker_context
from utils.load_model import WhisperForConditionalGeneration
from utils.model_utils import projection_module
from utils.reader import CustomDataset, write_jsonlines
from utils.utils import add_arguments, print_arguments


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    add_arg = functools.partial(add_arguments, argparser=parser)
    add_arg("test_data", type=str, default="dataset/test_data.jsonl", help="test list (jsonl)")
    add_arg("model_path", type=str, default="models/whisper-tiny-finetune", help="merged model dir or synthetic:<size>")
    add_arg("lora_model", type=str, default=None, help="trained adapter directory (results are written there)")
    add_arg("modal", type=str, default="speech", help="input modality")
    add_arg("sampling_rate", type=int, default=1000, help="signal sample rate")
    add_arg("eeg_ch", type=int, default=66, help="input channels")
    add_arg("batch_size", type=int, default=16, help="decode batch size")
    add_arg("num_workers", type=int, default=8, help="data loader workers")
    add_arg("language", type=str, default="Chinese", help="language")
    add_arg("remove_pun", type=bool, default=True, help="strip punctuation")
    add_arg("to_simple", type=bool, default=True, help="to simplified Chinese")
    add_arg("timestamps", type=bool, default=True, help="timestamp labels")
    add_arg("min_audio_len", type=float, default=0.5, help="minimum length (s)")
    add_arg("max_audio_len", type=float, default=30, help="maximum length (s)")
    add_arg("local_files_only", type=bool, default=True, help="never download")
    add_arg("noise", type=bool, default=False, help="feed noise instead of the signal")
    add_arg("filter_dataset", type=bool, default=False, help="filter the data list")
    add_arg("random_choice", type=bool, default=False, help="random label baseline")
    add_arg("task", type=str, default="transcribe", choices=["transcribe", "translate"], help="task")
    add_arg("random_initialize_whisper", type=bool, default=False, help="random init")
    add_arg("teacher_forcing", type=bool, default=False, help="teacher-forced argmax instead of generate")
    add_arg("extra_name", type=str, default=None, help="suffix for the result basename")
    add_arg("post_processing", type=bool, default=False, help="ascii/lowercase post-processing")
    add_arg("config_name", type=str, default="base", help="front-end module")
    add_arg("add_sequence_bias", type=bool, default=False, help="sequence bias")
    # additions of this build
    add_arg("num_beams", type=int, default=5, help="beam width (reference: 5)")
    add_arg("max_new_tokens", type=int, default=None, help="cap on generated tokens")
    add_arg("device_feed", type=bool, default=True, help="slice / pad / cast the recordings on the GPU (ns_feed_pack)")
    add_arg("feed_cache_dir", type=str, default="", help="with --device_feed: narrow-type cache of the kept channel rows (neuspeech1_amd/feed.py)")
    add_arg("feed_cache_dtype", type=str, default="f16", help="f16 | f32 (see --feed_cache_dir)")
    add_arg("sequence_bias_type", type=str, default="phrase_word", choices=["word", "phrase", "phrase_word"],
            help="what --add_sequence_bias extracts from the training sentences (reference: phrase_word, needs yake)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.num_workers > 0:
        start_worker_server()       # the DataLoader workers' forkserver: a fresh helper process, started before anything touches the GPU
    print_arguments(args)
    assert args.model_path.startswith("synthetic:") or os.path.exists(args.model_path), f"model {args.model_path} not found"
    from finetune import get_processor
    processor = get_processor(args.model_path, args.language, args.task, args.timestamps, args.local_files_only)
    # one process per GPU under torchrun: every rank is a full replica decoding a strided shard of the test list; the only
    # exchange is the gather of the decoded TEXT at the end ("replicas only").  Single process otherwise,
    # exactly the reference's flow.
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK") or 0)
    device_map = "auto"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            local %= max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
        device_map = {"": local}
        dist.init_process_group("gloo")         # text gather only: no GPU collective on the decode path
    model = WhisperForConditionalGeneration.from_pretrained(args.model_path, device_map=device_map,
                                                            local_files_only=args.local_files_only)
    torch.manual_seed(42)
    loaded = model.model.encoder.conv1
    first = loaded[0] if isinstance(loaded, torch.nn.Sequential) else loaded
    merged_export = args.lora_model is None and first.in_channels == args.eeg_ch and \
        (isinstance(loaded, torch.nn.Sequential) or loaded.stride == (2,))
    if not merged_export:     # a merged export (merge_lora.py) already carries its trained front-end: keep it
        conv1 = projection_module(config_name=args.config_name, meg_ch=args.eeg_ch,
                                  d_model=model.model.encoder.conv2.in_channels).to(model.device)
        model.model.encoder.set_input_embeddings(conv1)
    out_dir = args.lora_model or "."
    if args.lora_model is not None:
        model = PeftModel.from_pretrained(model, args.lora_model, local_files_only=args.local_files_only)
        model = model.merge_and_unload()
    if args.random_initialize_whisper:
        model.model.decoder.post_init()     # reference :90-91: the decoder only
        model._engine = None
    model.eval()
    test_dataset = CustomDataset(data_list_path=args.test_data, processor=processor, timestamps=args.timestamps,
                                 modal=args.modal, mode="test", modal_ch=args.eeg_ch, filter_dataset=args.filter_dataset,
                                 sample_rate=args.sampling_rate, language=args.language,
                                 min_duration=args.min_audio_len, max_duration=args.max_audio_len)
    print(f"test samples: {len(test_dataset)}")
    collator = DataCollatorSpeechSeq2SeqWithPadding(processor=processor)
    shard = test_dataset if world == 1 else torch.utils.data.Subset(test_dataset, list(range(rank, len(test_dataset), world)))
    loader = torch.utils.data.DataLoader(shard, batch_size=args.batch_size, num_workers=args.num_workers, collate_fn=collator,
                                         multiprocessing_context=worker_context(args.num_workers))
    base = (f'formal_test_results{"_" + args.extra_name if args.extra_name is not None else ""}'
            f'{"no_post_processing" if not args.post_processing else "post_processing"}'
            f'{"_noise" if args.noise else ""}{"_randomChoice" if args.random_choice else ""}'
            f'{"_tf" if args.teacher_forcing else ""}')
    if args.random_choice:
        assert world == 1, "--random_choice is a single-process baseline"
        # chance baseline of the reference (:330-331, :406-420, :462-466): every prediction is a label drawn at random
        # from the test list itself; the model is not run
        all_labels = []
        for batch in fork_safe_iter(loader):
            lab = np.where(batch["labels"].numpy() != -100, batch["labels"].numpy(), processor.tokenizer.pad_token_id)
            all_labels.extend(processor.batch_decode(lab, skip_special_tokens=True))
        all_preds = np.random.choice(all_labels, len(all_labels)).tolist()
        write_jsonlines(os.path.join(out_dir, base + ".jsonl"), [{"pred": p, "label": l} for p, l in zip(all_preds, all_labels)])
        results = {"samples": len(all_labels), "random_choice": True}
        print(f"results: {results}")
        with open(os.path.join(out_dir, base + ".json"), "w") as f:
            json.dump(results, f)
        return
    feed = None
    if args.device_feed and not args.noise and model.device.type == "cuda":
        # recordings go to the GPU as bytes; slice / pad / crop / cast run there (neuspeech1_amd/feed.py)
        from neuspeech1_amd.feed import SignalFeed
        dims = model.engine().dims
        test_dataset.raw_signals = True
        feed = SignalFeed(model.device, dims.ch, dims.T, dims.ch_pad, threads=max(2, args.num_workers),
                          cache_dir=args.feed_cache_dir or None, cache_dtype=args.feed_cache_dtype)

    def batches():
        """(input, labels) with the NEXT batch's file reads already running on the feed's loader thread"""
        it = fork_safe_iter(loader)
        stage = lambda b: feed.submit(b["input_features"]) if (feed is not None and b is not None) else None  # noqa: E731
        nxt = next(it, None)
        fut = stage(nxt)
        while nxt is not None:
            cur, cur_fut = nxt, fut
            nxt = next(it, None)
            fut = stage(nxt)
            yield (cur_fut.result() if cur_fut is not None else cur["input_features"].to(model.device)), cur["labels"]

    sequence_bias = None
    if args.add_sequence_bias:      # reference :339-343: bias -1.0 on the words / key phrases of the training list
        from utils.generation_helper import GetSequenceBias
        tok = processor.tokenizer if args.model_path.startswith("synthetic:") else None
        sequence_bias = GetSequenceBias(tokenizer_name=args.model_path, jsonl_path=args.test_data.replace("test.jsonl", "train.jsonl"),
                                        bias=-1.0, extract_type=args.sequence_bias_type, tokenizer=tok).get_bias_for_my_sentences()
        print(f"sequence bias: {len(sequence_bias)} token sequences")
    preds, refs, tf_ids = [], [], []
    n_new, n_match, n_lab, t0 = 0, 0, 0, time.time()
    t_gen = 0.0
    txt_path = os.path.join(out_dir, base + ".txt") if world == 1 else os.devnull   # ranks > 1: rank 0 writes it after the gather
    with open(txt_path, "w") as f, torch.no_grad():
        for x, labels in batches():
            if args.noise:
                x = torch.randn_like(x)
            if not args.teacher_forcing:
                kw = {}
                if args.language.lower() != "english":
                    kw["decoder_input_ids"] = labels[:, :4].to(model.device)
                if args.max_new_tokens is not None:
                    kw["max_new_tokens"] = args.max_new_tokens
                if sequence_bias is not None:
                    kw["sequence_bias"] = sequence_bias
                tg = time.time()
                gen = model.generate(x, do_sample=False, num_beams=args.num_beams, repetition_penalty=5.0,
                                     no_repeat_ngram_size=2, **kw).cpu().numpy()
                t_gen += time.time() - tg
                n_new += int(gen.shape[0] * (gen.shape[1] - (4 if kw.get("decoder_input_ids") is not None else 1)))
            else:
                ign = labels == -100
                fed = labels.masked_fill(ign, model.config.eos_token_id)
                logits = model(input_features=x, decoder_input_ids=fed.to(model.device)).logits
                gen = logits.argmax(-1).cpu()
                n_match += int(((gen[:, :-1] == labels[:, 1:]) & ~ign[:, 1:]).sum())
                n_lab += int((~ign[:, 1:]).sum())
                gen = gen.masked_fill(ign, -100).numpy()
                tf_ids.append(gen)
            if hasattr(x, "release"):
                x.release()
            lab = np.where(labels.numpy() != -100, labels.numpy(), processor.tokenizer.pad_token_id)
            dp = processor.batch_decode(gen, skip_special_tokens=True)
            dl = processor.batch_decode(lab, skip_special_tokens=True)
            preds.extend(dp)
            refs.extend(dl)
            if args.post_processing:    # reference :417-421: the .txt listing shows the filtered text, the .jsonl keeps the raw one
                from utils.process_str import convert_lower_text, filter_ascii_text
                dp, dl = convert_lower_text(filter_ascii_text(dp)), convert_lower_text(filter_ascii_text(dl))
            for p, l in zip(dp, dl):
                f.write("start********************************\n")
                f.write(f"Predicted: {p}\nTrue: {l}\n")
                f.write("end==================================\n\n")
    dt = time.time() - t0
    if feed is not None:
        feed.close()        # its loader / reader threads end here, not whenever the collector finds the feed
    if hasattr(model, "release_decode_sessions"):
        model.release_decode_sessions()     # the loop is over: the sessions' caches / graphs (tens of GB at large-v2) go back to the allocator
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (preds, refs, n_new, n_match, n_lab, dt, t_gen))
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
        n = sum(len(p[0]) for p in parts)
        preds = [parts[i % world][0][i // world] for i in range(n)]     # undo the strided shard: original list order
        refs = [parts[i % world][1][i // world] for i in range(n)]
        n_new, n_match, n_lab = (sum(p[k] for p in parts) for k in (2, 3, 4))
        dt, t_gen = max(p[5] for p in parts), max(p[6] for p in parts)
        with open(os.path.join(out_dir, base + ".txt"), "w") as f:
            dp, dl = preds, refs
            if args.post_processing:
                from utils.process_str import convert_lower_text, filter_ascii_text
                dp, dl = convert_lower_text(filter_ascii_text(dp)), convert_lower_text(filter_ascii_text(dl))
            for p, l in zip(dp, dl):
                f.write("start********************************\n")
                f.write(f"Predicted: {p}\nTrue: {l}\n")
                f.write("end==================================\n\n")
    write_jsonlines(os.path.join(out_dir, base + ".jsonl"), [{"pred": p, "label": l} for p, l in zip(preds, refs)])
    results = {"samples": len(preds), "n_gpus": world, "seconds": round(dt, 3), "generate_seconds": round(t_gen, 3),
               "generated_tokens_per_s": round(n_new / dt, 2) if n_new else None,
               "teacher_forced_token_accuracy": round(n_match / n_lab, 5) if n_lab else None}
    print(f"results: {results}")
    with open(os.path.join(out_dir, base + ".json"), "w") as f:
        json.dump(results, f)
    # callers in-process (tests) also get the model and, for --teacher_forcing, the argmax ids per batch (-100 on padding)
    return dict(results, model=model, teacher_forced_ids=tf_ids)


if __name__ == "__main__":
    main()
