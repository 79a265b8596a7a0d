"""Is the samples/s that finetune.py logs a GPU rate?  Wrap the engine's train_step with a device-wide synchronize
every 10 steps and compare with the CLI's own log."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import finetune
from neuspeech1_amd import engine as E
from neuspeech1_amd.synthetic import write_synthetic_dataset
orig = E.MegWhisperEngine.train_step
state = {"n": 0, "t": None, "rates": []}
def timed(self, x, y, **kw):
    if state["n"] % 10 == 0:
        torch.cuda.synchronize()
        now = time.perf_counter()
        if state["t"] is not None:
            state["rates"].append(round(640 / (now - state["t"]), 1))
        state["t"] = now
    state["n"] += 1
    return orig(self, x, y, **kw)
E.MegWhisperEngine.train_step = timed
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    jl = write_synthetic_dataset(os.path.join(tmp, "data"), 512, ch_file=224, name="gwilliams", seed=6, min_len=600, max_len=3000)
    rows = [l for l in open(jl)]
    with open(jl, "w") as f:
        for k in range(8):
            f.writelines(rows)
    out = os.path.join(tmp, "out")
    finetune.main(["--per_device_train_batch_size=64", "--per_device_eval_batch_size=64", f"--output_dir={out}", "--eval_steps=1000",
                   "--save_steps=1000", "--learning_rate=1e-3", "--fp16=True", "--num_train_epochs=500", "--warmup_steps=500",
                   "--num_workers=8", "--modal=eeg", "--eeg_ch=208", "--sampling_rate=200", "--orig_sample_rate=200",
                   f"--train_data={jl}", f"--test_data={jl}", "--base_model=synthetic:base", "--use_adalora=False",
                   "--augment_config_path=configs/augmentation1.json", "--language=English", "--device=cuda", "--logging_steps=10",
                   "--max_steps=60"] + sys.argv[1:])
    logs = [json.loads(l) for l in open(os.path.join(out, "synthetic_base", "train_log.jsonl"))]
print("CLI log   ", [l["samples_per_s"] for l in logs])
print("synced    ", state["rates"])
