"""The reference's Gwilliams training recipe (reference README.md:23-32: bs 64, fp16, default AdaLoRA, default timestamp
labels, configs/augmentation1.json) through finetune.py on a synthetic list at whisper-base size; prints the
samples/s the CLI itself logs.  python tools/run_recipe.py [steps]"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
extra = sys.argv[2:]
import finetune
from neuspeech1_amd.synthetic import write_synthetic_dataset
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    schof = any(a == "--eeg_ch=273" for a in extra)      # the Schoffelen recipe (README.md:43-56): rows [28:301] of 301+
    jl = write_synthetic_dataset(os.path.join(tmp, "data"), 512, ch_file=301 if schof else 224,
                                 name="schoffelen" if schof else "gwilliams", seed=6, min_len=600, max_len=3000)
    rows = [l for l in open(jl)]
    with open(jl, "w") as f:            # 4096 list entries over the 512 recordings: one epoch covers the whole run
        for k in range(8):
            f.writelines(rows)
    out = os.path.join(tmp, "out")
    finetune.main(["--per_device_train_batch_size=64", "--per_device_eval_batch_size=64", f"--output_dir={out}",
                   "--eval_steps=1000", "--save_steps=1000", "--learning_rate=1e-3", "--fp16=True", "--num_train_epochs=500",
                   "--warmup_steps=500", "--max_audio_len=30", "--use_8bit=False", "--num_workers=8", "--modal=eeg", "--eeg_ch=208",
                   "--sampling_rate=200", "--orig_sample_rate=200", f"--train_data={jl}", f"--test_data={jl}",
                   "--base_model=synthetic:base", "--augment_config_path=configs/augmentation1.json", "--language=English",
                   "--device=cuda", "--logging_steps=10", f"--max_steps={steps}"] + extra)
    logs = [json.loads(l) for l in open(os.path.join(out, "synthetic_base", "train_log.jsonl"))]
    print("RECIPE", json.dumps({"first_loss": logs[0]["loss"], "last_loss": logs[-1]["loss"],
                                "samples_per_s": [l["samples_per_s"] for l in logs]}))
