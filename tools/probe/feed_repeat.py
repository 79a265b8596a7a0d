"""run-to-run spread of three optimizer steps (tensor path twice, feed path twice), optionally without the fused kernels"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
from neuspeech1_amd.weights import TINY, make_state_dict, synth_batch
dev = torch.device("cuda:0")
dims = TINY
x, labels = synth_batch(dims, 4, 77)
x32 = torch.from_numpy(x).to(dev); labels = torch.from_numpy(labels).to(dev)
def run(nofuse):
    torch.manual_seed(3)
    eng = MegWhisperEngine(dims, make_state_dict(dims, 42), lora=LoraSpec(r=8, alpha=16.0, dropout=0.0),
                           train_cfg=TrainCfg(lr=1e-3, warmup_steps=2, total_steps=10), device=dev)
    eng.no_fused_lora_bwd = nofuse
    losses = [eng.train_step(x32, labels).item() for _ in range(3)]
    torch.cuda.synchronize()
    return losses, eng.P.clone(), eng
for nofuse in (False, True):
    runs = [run(nofuse) for _ in range(3)]
    for i in range(1, 3):
        d = (runs[0][1] - runs[i][1]).abs()
        j = int(d.argmax())
        eng = runs[0][2]
        name = [k for k, (o, n) in eng.seg_off.items() if o <= j < o + n]
        print("nofuse", nofuse, "losses", runs[0][0], runs[i][0], "max |dP|", d.max().item(), name, j - eng.seg_off[name[0]][0] if name else None)
