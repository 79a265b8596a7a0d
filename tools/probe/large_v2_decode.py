"""whisper-large-v2 dims through the decode loop (BASELINE configs[4]'s model): runs, tokens/s, greedy ids vs beam-1."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd.engine import MegWhisperEngine
from neuspeech1_amd.generate import Generator
from neuspeech1_amd.weights import WHISPER_LARGE_V2, WhisperDims, make_state_dict, synth_batch
import dataclasses
dev = torch.device("cuda:0")
dims = dataclasses.replace(WHISPER_LARGE_V2, ch=273)
t0 = time.time(); sd = make_state_dict(dims, 42); print(f"weights {time.time() - t0:.0f} s", flush=True)
eng = MegWhisperEngine(dims, sd, device=dev); del sd
gen = Generator(eng)
B, NEW = int(os.environ.get("B", 32)), 32
x, labels = synth_batch(dims, B, 1234)
x = torch.from_numpy(x).to(dev); prompt = torch.from_numpy(labels[:, :4].copy()).to(dev)
for nb, kw in ((1, {}), (5, dict(repetition_penalty=5.0, no_repeat_ngram_size=2))):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = gen.generate(x, prompt, num_beams=nb, max_new_tokens=NEW, suppress_tokens=[dims.eos_id], check_every=8, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"large-v2 beams={nb}: {B * (out.shape[1] - 4) / dt:.0f} tokens/s ({dt * 1e3:.0f} ms, B={B}), ids {out[0, 4:10].tolist()}", flush=True)
