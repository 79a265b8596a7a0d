"""--ft_full at whisper-base dims: a few training steps (loss must fall, no overflow) and the step time next to the
encoder-only adapters."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch
dev = torch.device("cuda:0")
dims = WhisperDims(ch=208)
B = int(os.environ.get("B", 64))
x, labels = synth_batch(dims, B, 1234)
xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
sd = make_state_dict(dims, 42)
for dec in (False, True):
    for ada in (False, True):
        torch.manual_seed(42)
        spec = LoraSpec(r=12 if ada else 32, alpha=32.0 if ada else 64.0, dropout=0.1 if ada else 0.05, adalora=ada,
                        orth_reg_weight=0.5 if ada else 0.0, decoder=dec)
        eng = MegWhisperEngine(dims, sd, lora=spec, train_cfg=TrainCfg(lr=1e-3, warmup_steps=0, total_steps=0, fp16_scaler=True), device=dev)
        ls = [eng.train_step(xd, ld).item() for _ in range(4)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): eng.train_step(xd, ld)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"decoder={dec} adalora={ada}: trainable {eng.n_train/1e6:.2f} M, losses {[round(v, 3) for v in ls]}, "
              f"{dt*1e3:.1f} ms/step, inf={eng.found_inf_dev.item()} scale={eng.loss_scale_dev.item():.0f}", flush=True)
        del eng
        torch.cuda.empty_cache()
