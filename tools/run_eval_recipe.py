"""evaluation.py end to end at BASELINE configs[3] scale (whisper-base, 273 channels, B = 128, beam 5 + repetition
penalty + no-repeat-2) on a synthetic list: the tokens/s the CLI itself reports.  python tools/run_eval_recipe.py"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import evaluation
from neuspeech1_amd.synthetic import write_synthetic_dataset
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    jl = write_synthetic_dataset(os.path.join(tmp, "data"), 256, ch_file=301, name="schoffelen", seed=8, min_len=600, max_len=3000)
    rows = [l for l in open(jl)]
    with open(jl, "w") as f:
        for k in range(int(os.environ.get("REP", 4))):
            f.writelines(rows)
    os.chdir(tmp)
    evaluation.main([f"--test_data={jl}", "--model_path=synthetic:base", "--modal=eeg", "--sampling_rate=200", "--eeg_ch=273",
                     "--batch_size=128", "--num_workers=8", "--language=Dutch", "--timestamps=False", "--max_new_tokens=64"] + sys.argv[1:])
    print("EVAL", open("formal_test_resultsno_post_processing.json").read())
