"""Soak: N optimizer steps of the BASELINE configs[1] workload on fresh synthetic batches (a small cycling pool), checking
that the loss falls, stays finite, and that the loss scaler never collapses.  Usage: python tools/soak_train.py [steps] [ch] [lora|adalora|full]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 208
kind = sys.argv[3] if len(sys.argv) > 3 else "lora"   # adalora: the reference's default adapter; full: --ft_full (decoder adapters too)
dev = torch.device("cuda:0")
dims = WhisperDims(ch=ch)
torch.manual_seed(42)
eng = MegWhisperEngine(dims, make_state_dict(dims, 42), lora=LoraSpec(r=32, alpha=64.0, dropout=0.05, adalora=(kind == "adalora"), decoder=(kind == "full")),
                       train_cfg=TrainCfg(lr=1e-3, warmup_steps=50, total_steps=steps), device=dev)
pool = []
for i in range(4):
    x, labels = synth_batch(dims, 64, 1000 + i, full_len=(i % 2 == 0))
    pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)))
t0 = time.time()
hist = []
for s in range(steps):
    x, y = pool[s % len(pool)]
    loss = eng.train_step(x, y)
    if s % 20 == 0 or s == steps - 1:
        torch.cuda.synchronize()
        hist.append((s, loss.item(), eng.loss_scale_dev.item(), int(eng.found_inf_dev.item())))
        print(f"step {s:4d} loss {hist[-1][1]:.4f} scale {hist[-1][2]:.0f} inf {hist[-1][3]} {64 * (s + 1) / (time.time() - t0):.0f} samples/s", flush=True)
assert all(torch.isfinite(torch.tensor(h[1])) for h in hist) and hist[-1][1] < hist[0][1] - 1.0, hist
assert hist[-1][2] >= 1024, "loss scale collapsed"
print("soak ok")
