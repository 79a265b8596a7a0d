"""Race screen for the decode kernels: the same generation repeated N times must give bit-identical ids and scores
(small-M split-K GEMM, one-pass attention, few-query MFMA attention, fused selection, K/V append are all
deterministic by construction: fixed reduction orders, no atomics)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd.engine import MegWhisperEngine
from neuspeech1_amd.generate import Generator
from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch
dev = torch.device("cuda:0")
dims = WhisperDims(ch=208)
gen = Generator(MegWhisperEngine(dims, make_state_dict(dims, 42), device=dev))
B, N = 24, int(os.environ.get("N", 30))
x, labels = synth_batch(dims, B, 77, full_len=False)
x = torch.from_numpy(x).to(dev); prompt = torch.from_numpy(labels[:, :4].copy()).to(dev)
bad = 0
for nb, kw in ((1, dict(repetition_penalty=5.0, no_repeat_ngram_size=2)), (5, dict(repetition_penalty=5.0, no_repeat_ngram_size=2)), (5, {})):
    ref = ref_s = None
    for it in range(N):
        out = gen.generate(x, prompt, num_beams=nb, max_new_tokens=40, check_every=4, **kw)
        sc = gen.last_scores.clone() if nb > 1 else None
        if ref is None:
            ref, ref_s = out.clone(), sc
        elif not torch.equal(out, ref) or (sc is not None and not torch.equal(sc, ref_s)):
            bad += 1
    print(f"beams={nb} {kw}: {N} runs, mismatching runs so far {bad}", flush=True)
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
