import sys, time, torch
sys.path.insert(0, "/root/repo")
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
from neuspeech1_amd.weights import WHISPER_LARGE_V2, make_state_dict, synth_batch
dev = torch.device("cuda:0")
dims = WHISPER_LARGE_V2
t0 = time.time()
sd = make_state_dict(dims, 42)
print("weights", time.time() - t0, flush=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = MegWhisperEngine(dims, sd, lora=LoraSpec(r=32, alpha=64.0, dropout=0.05), train_cfg=TrainCfg(lr=1e-4, warmup_steps=0, total_steps=0), device=dev)
del sd
x, labels = synth_batch(dims, B, 1234)
x, labels = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
losses = []
for i in range(6):
    torch.cuda.synchronize(); t = time.time()
    l = eng.train_step(x, labels)
    torch.cuda.synchronize(); dt = time.time() - t
    losses.append(l.item())
    print(f"step {i} loss {losses[-1]:.4f} {dt*1e3:.1f} ms  {B/dt:.1f} samples/s found_inf {eng.found_inf_dev.item()} scale {eng.loss_scale_dev.item()}", flush=True)
print("mem GB", torch.cuda.max_memory_allocated() / 2**30)
