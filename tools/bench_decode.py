"""Decode throughput at BASELINE configs[3]: whisper-base, 273-ch, B=128, beam 5 + repetition penalty 5.0 + no-repeat-2,
64 new tokens with EOS suppressed (every row does identical work).  tokens/s = emitted tokens (prompt excluded) summed over
the batch / wall time including the encoder pass."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuspeech1_amd.engine import MegWhisperEngine  # noqa: E402
from neuspeech1_amd.generate import Generator  # noqa: E402
from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch  # noqa: E402

if "NS_RING" in os.environ:        # kernel-choice A/B (ns_debug_set_ring, see include/neuspeech_hip.h)
    from neuspeech1_amd import lib as _l
    _l.load().ns_debug_set_ring(int(os.environ["NS_RING"]))
B = int(os.environ.get("B", 128))
NEW = int(os.environ.get("NEW", 64))
dev = torch.device("cuda:0")
dims = WhisperDims(ch=273)
eng = MegWhisperEngine(dims, make_state_dict(dims, 42), device=dev)
gen = Generator(eng, use_graph=os.environ.get("GRAPH", "1") == "1")
x, labels = synth_batch(dims, B, 1234)
x = torch.from_numpy(x).to(dev)
prompt = torch.from_numpy(labels[:, :4].copy()).to(dev)
ONLY = os.environ.get("BEAMS")     # "1" or "5": one mode only (clean rocprofv3 kernel stats per mode)
SB = {}
if os.environ.get("SB"):           # evaluation.py --add_sequence_bias: a small table of single- and multi-token entries
    SB = dict(sequence_bias={(7,): 2.0, (11, 12): 1.5, (20, 21, 22): 3.0, (300,): -1.0, (41, 42): 0.5})
MARK = os.environ.get("MARK") == "1"     # tools/profile.sh decode_pmc: one run per (mode, length), a marker launch between them


def mark():
    """a launch no decode step issues (torch's scan kernel): tools/pmc_decode_summary.py cuts the dispatch list at it"""
    torch.cumsum(torch.ones(64, device=dev), 0)


if MARK:
    modes = ((1, {}), (5, dict(repetition_penalty=5.0, no_repeat_ngram_size=2)))
    for nb, kw in modes:         # warm-up: lazy allocations (and their fills) stay out of the counted runs
        gen.generate(x, prompt, num_beams=nb, max_new_tokens=8, suppress_tokens=[dims.eos_id], check_every=8, **kw)
    torch.cuda.synchronize()
    mark()
    for nb, kw in modes:
        for new in (NEW, NEW // 2):
            out = gen.generate(x, prompt, num_beams=nb, max_new_tokens=new, suppress_tokens=[dims.eos_id], check_every=8, **kw)
            torch.cuda.synchronize()
            mark()
            print(f"segment beams={nb} new={out.shape[1] - 4}", flush=True)
    sys.exit(0)
for nb, kw in ((1, dict(SB)), (5, dict(repetition_penalty=5.0, no_repeat_ngram_size=2, **SB))):
    if ONLY and int(ONLY) != nb:
        continue
    for it in range(3):     # the third call of a signature is the session's steady state (lists, then the capture, then replays)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gen.generate(x, prompt, num_beams=nb, max_new_tokens=NEW, suppress_tokens=[dims.eos_id], check_every=8, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n = B * (out.shape[1] - 4)
    print(f"beams={nb}: {n} tokens in {dt * 1e3:.1f} ms -> {n / dt:.0f} tokens/s ({dt / (out.shape[1] - 4) * 1e3:.2f} ms/step incl. encoder; loop: {gen.last_loop_mode})", flush=True)
