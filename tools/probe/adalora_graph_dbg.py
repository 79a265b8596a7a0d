import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 273
npool = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
dims = WhisperDims(ch=ch)
torch.manual_seed(42)
eng = MegWhisperEngine(dims, make_state_dict(dims, 42), lora=LoraSpec(r=32, alpha=64.0, dropout=0.05, adalora=True),
                       train_cfg=TrainCfg(lr=1e-3, warmup_steps=50, total_steps=100), device=dev)
pool = []
for i in range(npool):
    x, labels = synth_batch(dims, 64, 1000 + i, full_len=(i % 2 == 0))
    pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)))
    print("batch", i, "labels", tuple(labels.shape))
for s in range(60):
    x, y = pool[s % len(pool)]
    loss = eng.train_step(x, y)
    if s % 2 == 0 or s > 20:
        torch.cuda.synchronize()
        print(s, "ret", float(loss.item()), "loss_dev", eng.loss_dev.item(), "reg", eng.reg_dev.item(), "tot", eng.total_loss_dev.item(),
              "graphs", len(eng._graphs), "P finite", bool(torch.isfinite(eng.P).all()), "|P|", eng.P.norm().item(), flush=True)
    if not torch.isfinite(loss).all() or loss.item() > 1e6:
        break
