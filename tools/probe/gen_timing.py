"""wall-clock split of one generate() call (B = 128, 273-ch, 64 new tokens): encoder / cross K,V / prompt / loop / tail"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuspeech1_amd.engine import MegWhisperEngine
from neuspeech1_amd import generate as G
from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch
dev = torch.device("cuda:0")
dims = WhisperDims(ch=273)
eng = MegWhisperEngine(dims, make_state_dict(dims, 42), device=dev)
x, labels = synth_batch(dims, 128, 1234)
x = torch.from_numpy(x).to(dev); prompt = torch.from_numpy(labels[:, :4].copy()).to(dev)
marks = []
orig_encode = eng.encode
def enc(*a, **k):
    torch.cuda.synchronize(); marks.append(("enc_begin", time.perf_counter()))
    r = orig_encode(*a, **k)
    torch.cuda.synchronize(); marks.append(("enc_end", time.perf_counter()))
    return r
eng.encode = enc
for nb, kw in ((1, {}), (5, dict(repetition_penalty=5.0, no_repeat_ngram_size=2))):
    for use_graph in (True, False):
        gen = G.Generator(eng, use_graph=use_graph)
        for it in range(2):
            marks.clear()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = gen.generate(x, prompt, num_beams=nb, max_new_tokens=64, suppress_tokens=[dims.eos_id], check_every=8, **kw)
            torch.cuda.synchronize(); t1 = time.perf_counter()
        eb = [t for n, t in marks if n == "enc_begin"][0]; ee = [t for n, t in marks if n == "enc_end"][0]
        print(f"beams={nb} graph={use_graph}: total {1e3*(t1-t0):.1f} ms = before-encoder {1e3*(eb-t0):.1f} + encoder {1e3*(ee-eb):.1f} + rest {1e3*(t1-ee):.1f}  "
              f"-> {128*(out.shape[1]-4)/(t1-t0):.0f} tok/s", flush=True)
