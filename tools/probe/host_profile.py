import os, sys, time, torch, cProfile, pstats
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
pr = cProfile.Profile()
pr.enable()
for _ in range(5): eng.train_step(x, labels)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
