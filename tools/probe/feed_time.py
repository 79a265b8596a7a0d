"""Where a file-fed training iteration spends host time: feed.load vs train_step enqueue vs GPU time per step."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
from neuspeech1_amd.feed import RawSignal, SignalFeed
from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch
dev = torch.device("cuda:0")
dims = WhisperDims(ch=208)
B, N = 64, int(os.environ.get("N", 6000))
mode = os.environ.get("MODE", "feed")
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    base = np.random.default_rng(0).standard_normal((224, N))
    raws = []
    for i in range(128):
        p = os.path.join(tmp, f"g{i}.npy"); np.save(p, base + i); raws.append(RawSignal(p, 0, 208, 208))
    torch.manual_seed(42)
    eng = MegWhisperEngine(dims, make_state_dict(dims, 42), lora=LoraSpec(r=32, alpha=64.0, dropout=0.05),
                           train_cfg=TrainCfg(lr=1e-3, warmup_steps=500, total_steps=100000), device=dev)
    x0, labels = synth_batch(dims, B, 1)
    y = torch.from_numpy(labels).to(dev); x0 = torch.from_numpy(x0).to(dev)
    feed = SignalFeed(dev, dims.ch, dims.T, dims.ch_pad, threads=int(os.environ.get("THREADS", 8)))
    if mode in ("prefetch", "prefetch_host"):
        from finetune import DevicePrefetcher
        ycpu = torch.from_numpy(labels)
        xcpu = torch.from_numpy(synth_batch(dims, B, 1)[0]).pin_memory()
        bt_shape = []
        def gen():
            if os.environ.get("DL"):
                from neuspeech1_amd.synthetic import SyntheticProcessor
                from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding
                from utils.reader import CustomDataset
                proc = SyntheticProcessor(dims)
                jl = os.path.join(tmp, "l.jsonl")
                with open(jl, "w") as f:
                    for i, r in enumerate(raws):
                        f.write(json.dumps({"eeg": {"path": r.path.replace("/g", "/gwilliams") if False else r.path}, "sentence": f"sample number {i} of the feed bench", "language": "English", "duration": 30}) + "\n")
                ds = CustomDataset(data_list_path=jl, processor=proc, modal="eeg", modal_ch=208, mode="train", sample_rate=200,
                                   orig_sample_rate=200, language="English", timestamps=False, raw_signals=True)
                idx = [i % 128 for i in range(64 * 23)]
                dl = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, idx), batch_size=64, shuffle=False,
                                                 num_workers=int(os.environ["DL"]), collate_fn=DataCollatorSpeechSeq2SeqWithPadding(processor=proc))
                t_dl = 0.0
                it = iter(dl)
                while True:
                    a_ = time.perf_counter()
                    bt = next(it, None)
                    t_dl += time.perf_counter() - a_
                    if bt is None:
                        break
                    bt_shape[:] = list(bt["labels"].shape)
                    yield bt
                print("dataloader next() total ms per batch", round(t_dl / 23 * 1e3, 2), "label shape", tuple(bt_shape))
                return
            for k in range(23):
                if mode == "prefetch":
                    yield {"input_features": raws[(k % 2) * 64:(k % 2) * 64 + 64], "labels": ycpu}
                else:
                    yield {"input_features": xcpu, "labels": ycpu}
        tl = ts = 0.0
        k = 0
        a = time.perf_counter()
        for x, yy in DevicePrefetcher(gen(), dev, feed):
            if k == 3:
                torch.cuda.synchronize(); t0 = time.perf_counter(); tl = ts = 0.0
            b = time.perf_counter()
            eng.train_step(x, yy)
            if hasattr(x, "release"):
                x.release()
            c = time.perf_counter()
            tl += b - a; ts += c - b
            a = c
            k += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"mode": mode, "N": N, "ms_per_iter": round(dt / 20 * 1e3, 2), "host_load_ms": round(tl / 20 * 1e3, 2),
                          "host_step_ms": round(ts / 20 * 1e3, 2), "slots": len(feed.slots)}), flush=True)
        sys.exit(0)
    tl = ts = 0.0
    for k in range(23):
        if k == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter(); tl = ts = 0.0
        a = time.perf_counter()
        if mode == "feed":
            x = feed.load(raws[(k % 2) * 64:(k % 2) * 64 + 64])
        elif mode == "loadonly":
            x = feed.load(raws[(k % 2) * 64:(k % 2) * 64 + 64]); x.release(); x = x0
        else:
            x = x0
        b = time.perf_counter()
        eng.train_step(x, y)
        if mode == "feed":
            x.release()
        c = time.perf_counter()
        tl += b - a; ts += c - b
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"mode": mode, "N": N, "ms_per_iter": round(dt / 20 * 1e3, 2), "host_load_ms": round(tl / 20 * 1e3, 2),
                      "host_step_ms": round(ts / 20 * 1e3, 2)}), flush=True)
