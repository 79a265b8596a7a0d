"""Data-feed measurement on the GPU box: batches/s of the host reader + collator path against
the on-GPU feed (neuspeech1_amd/feed.py), the ns_feed_pack kernel's own rate, and the end-to-end training rate when
every step pulls a fresh batch from files (page-cache resident) either way.

  python tools/bench_feed.py [--batch 64] [--ch 208] [--n 6000] [--steps 20] [--workers 8]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--ch", type=int, default=208)
    ap.add_argument("--n", type=int, default=6000, help="samples per recording (<= 6000 means no crop)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--files", type=int, default=256)
    ap.add_argument("--only", type=str, default="", help="comma list of legs to run in THIS process")
    ap.add_argument("--data-dir", type=str, default="", help="reuse recordings written by a parent run")
    args = ap.parse_args()
    if not args.only:
        # one child process per leg (legs sharing a process disturb each other: pinned-memory pools, loader workers);
        # the parent never touches the GPU
        import subprocess
        merged = {}
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
            for leg in ("host_path_1proc", f"host_path_{args.workers}workers", "gpu_feed", "gpu_feed_cache_f32", "gpu_feed_cache_f16",
                        "train_host_path", "train_gpu_feed", "train_gpu_feed_cache_f16"):
                cmd = [sys.executable, os.path.abspath(__file__), "--only", leg, "--data-dir", tmp, "--batch", str(args.batch),
                       "--ch", str(args.ch), "--n", str(args.n), "--steps", str(args.steps), "--workers", str(args.workers),
                       "--files", str(args.files)]
                r = subprocess.run(cmd, capture_output=True, text=True)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not line:
                    print(r.stdout[-2000:], r.stderr[-2000:], file=sys.stderr)
                    raise SystemExit(f"leg {leg} failed")
                merged.update(json.loads(line[-1]))
        print(json.dumps(merged), flush=True)
        return
    from finetune import DevicePrefetcher
    from neuspeech1_amd import ops
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    from neuspeech1_amd.feed import SignalFeed
    from neuspeech1_amd.synthetic import SyntheticProcessor
    from neuspeech1_amd.weights import WhisperDims, make_state_dict
    from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding, worker_context
    from utils.reader import CustomDataset
    dev = torch.device("cuda:0")
    dims = WhisperDims(ch=args.ch)
    proc = SyntheticProcessor(dims)
    out = {"batch": args.batch, "ch": args.ch, "n": args.n, "workers": args.workers}
    import contextlib
    holder = contextlib.nullcontext(args.data_dir) if args.data_dir else \
        tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    with holder as tmp:
        rng = np.random.default_rng(0)
        rows = []
        base = rng.standard_normal((args.ch + 16, args.n))
        for i in range(args.files):
            p = os.path.join(tmp, f"gwilliams_{i}.npy") if args.ch == 208 else os.path.join(tmp, f"rec_{i}.npy")
            if not os.path.exists(p):
                np.save(p, base + i)
            rows.append({"eeg": {"path": p}, "sentence": f"sample number {i} of the feed bench", "language": "English",
                         "duration": args.n / 200})
        jl = os.path.join(tmp, "list.jsonl")
        with open(jl, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
        kw = dict(data_list_path=jl, processor=proc, modal="eeg", modal_ch=args.ch, mode="train", sample_rate=200,
                  orig_sample_rate=200, language="English", timestamps=False, min_duration=0.5, max_duration=30)
        coll = DataCollatorSpeechSeq2SeqWithPadding(processor=proc)
        out["file_MB"] = round(os.path.getsize(rows[0]["eeg"]["path"]) / 1e6, 2)

        def loader(raw, workers, steps=None):
            ds = CustomDataset(raw_signals=raw, **kw)
            idx = [i % args.files for i in range(args.batch * ((steps or args.steps) + 4))]
            return torch.utils.data.DataLoader(torch.utils.data.Subset(ds, idx), batch_size=args.batch, shuffle=False,
                                               num_workers=workers, collate_fn=coll, pin_memory=not raw,
                                               multiprocessing_context=worker_context(workers))

        # 1. producer rates alone (no training): host collator path, 0 and W workers; feed path
        for name, raw, w in (("host_path_1proc", False, 0), (f"host_path_{args.workers}workers", False, args.workers),
                             ("gpu_feed", True, 0), ("gpu_feed_cache_f32", True, 0), ("gpu_feed_cache_f16", True, 0)):
            if args.only and name not in args.only.split(","):
                continue
            cdt = name.rsplit("_", 1)[1] if "cache" in name else None
            feed = SignalFeed(dev, dims.ch, dims.T, dims.ch_pad, threads=args.workers,
                              cache_dir=os.path.join(tmp, "cache_" + cdt) if cdt else None, cache_dtype=cdt or "f16") if raw else None
            if cdt:     # build the cache outside the timed region (first touch of every file): the steady state is what a run sees
                from neuspeech1_amd.feed import RawSignal, plan_cached
                t0 = time.perf_counter()
                for r in rows:
                    plan_cached(RawSignal(r["eeg"]["path"], 0, args.ch, args.ch), dims.T, feed.cache_dir, cdt)
                out[name + "_build_s"] = round(time.perf_counter() - t0, 2)
            it = iter(DevicePrefetcher(loader(raw, w, 4 if (w == 0 and not raw) else None), dev, feed))
            for _ in range(3):
                x, _y = next(it)
                if raw:
                    x.release()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = 0
            for x, _y in it:
                if raw:
                    x.release()
                k += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name + "_batches_per_s"] = round(k / dt, 2)
            out[name + "_samples_per_s"] = round(k * args.batch / dt, 1)
            if raw:
                out[name + "_staged_MB_per_batch"] = round(feed.bytes_staged / (k + 3) / 1e6, 1)
            if raw and not cdt:
                # kernel alone: re-run ns_feed_pack on the last staged slot
                s = feed.slots[0]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    ops.feed_pack(s.items_dev, args.batch, dims.ch, dims.T, dims.ch_pad, s.xin, None)
                e0.record()
                for _ in range(20):
                    ops.feed_pack(s.items_dev, args.batch, dims.ch, dims.T, dims.ch_pad, s.xin, None)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                byts = args.batch * (min(args.n, dims.T) * dims.ch * 8 + (dims.T + 2) * dims.ch_pad * 2)
                out["feed_pack_ms"] = round(ms, 4)
                out["feed_pack_GBps"] = round(byts / ms / 1e6, 1)
            if raw:
                feed.close()

        # 2. training fed from files, both ways
        for name, raw in (("train_host_path", False), ("train_gpu_feed", True), ("train_gpu_feed_cache_f16", True)):
            if args.only and name not in args.only.split(","):
                continue
            cdt = "f16" if "cache" in name else None
            torch.manual_seed(42)
            eng = MegWhisperEngine(dims, make_state_dict(dims, 42), lora=LoraSpec(r=32, alpha=64.0, dropout=0.05),
                                   train_cfg=TrainCfg(lr=1e-3, warmup_steps=500, total_steps=100000), device=dev)
            feed = SignalFeed(dev, dims.ch, dims.T, dims.ch_pad, threads=args.workers,
                              cache_dir=os.path.join(tmp, "cache_" + cdt) if cdt else None, cache_dtype=cdt or "f16") if raw else None
            it = iter(DevicePrefetcher(loader(raw, args.workers), dev, feed))
            k = 0
            for x, y in it:
                if k == 3:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                eng.train_step(x, y)
                if raw:
                    x.release()
                k += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name + "_samples_per_s"] = round((k - 3) * args.batch / dt, 1)
            del eng, feed
            torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
