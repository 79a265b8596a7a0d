"""The decode loop on a SLOW HOST, simulated: every replayed launch of a launch list spins SPIN_US microseconds on the host first (the
runtime's own launch path on the pool's slower hosts costs ~13 us per launch: 1.00 ms per greedy step against 0.86 ms of GPU time).
With NS_DECODE_ADAPT the loop notices (two polled chunks in a row spend > 95 % of their wall time inside the replays) and captures
hipGraphs mid-generation.  Prints tokens/s and the loop mode for both settings, greedy and beam-5 (bench.py's eval workload)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuspeech1_amd import ops  # noqa: E402
from neuspeech1_amd.engine import MegWhisperEngine  # noqa: E402
from neuspeech1_amd.generate import Generator  # noqa: E402
from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch  # noqa: E402

SPINS = [float(v) * 1e-6 for v in os.environ.get("SPIN_US", "8,14").split(",")]
SPIN = 0.0
_orig = ops.LaunchList.replay


def slow_replay(self):
    st = ops._stream()
    for fn, name, args in self.calls:
        t = time.perf_counter() + SPIN
        while time.perf_counter() < t:
            pass
        rc = fn(*args, st)
        if rc:
            ops.L.check(rc, name)


dev = torch.device("cuda:0")
dims = WhisperDims(ch=273)
eng = MegWhisperEngine(dims, make_state_dict(dims, 42), device=dev)
x, labels = synth_batch(dims, 128, 1234)
x = torch.from_numpy(x).to(dev)
prompt = torch.from_numpy(labels[:, :4].copy()).to(dev)
for spin in [0.0] + SPINS:
    SPIN = spin
    host = "as is" if spin == 0.0 else f"+{SPIN * 1e6:.0f} us per replayed launch"
    ops.LaunchList.replay = _orig if spin == 0.0 else slow_replay
    for nb, kw in ((1, {}), (5, dict(repetition_penalty=5.0, no_repeat_ngram_size=2))):
        for adapt in (False, True):
            gen = Generator(eng)
            gen.adaptive = adapt
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = gen.generate(x, prompt, num_beams=nb, max_new_tokens=64, suppress_tokens=[dims.eos_id], check_every=8, **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print(f"host {host:32s} beams {nb} adaptive {adapt!s:5s}: {128 * 64 / best:9.0f} tokens/s  ({best * 1e3:6.1f} ms)  loop: {gen.last_loop_mode}", flush=True)
