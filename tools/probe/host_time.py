import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch
dev = torch.device("cuda:0")
dims = WhisperDims(ch=208)
eng = MegWhisperEngine(dims, make_state_dict(dims, 42), lora=LoraSpec(r=32, alpha=64.0, dropout=0.05), train_cfg=TrainCfg(), device=dev)
x, labels = synth_batch(dims, 64, 1234)
x, labels = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
for _ in range(3): eng.train_step(x, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): eng.train_step(x, labels)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step, cpus {os.cpu_count()}")
