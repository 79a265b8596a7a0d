"""Synthetic recordings shared by the feed tests (CPU: host planning / staging; GPU: ns_feed_pack)."""
import json
import os

import numpy as np


def write_cases(root, modal_ch):
    """Recordings that exercise every reader rule and every staging path; returns the jsonl path."""
    rng = np.random.default_rng(modal_ch)
    os.makedirs(os.path.join(root, "gwilliams"), exist_ok=True)
    os.makedirs(os.path.join(root, "schoffelen"), exist_ok=True)
    os.makedirs(os.path.join(root, "other"), exist_ok=True)
    specs = [
        ("gwilliams/short_f64.npy", rng.standard_normal((224, 700))),                    # [:208], time pad
        ("gwilliams/exact_f32.npy", rng.standard_normal((208, 6000)).astype(np.float32)),
        ("gwilliams/crop_f64.npy", rng.standard_normal((230, 7321))),                    # span read, cropped on the GPU
        ("gwilliams/long_f16.npy", rng.standard_normal((208, 12500)).astype(np.float16)),  # n > 2T: row-wise reads
        ("schoffelen/s_f64.npy", rng.standard_normal((301, 1999)) * 1e3),                # [28:301] = 273 rows
        ("other/few_rows.npy", rng.standard_normal((100, 3001))),                        # zero channels appended
        ("other/fortran.npy", np.asfortranarray(rng.standard_normal((modal_ch, 900)))),  # host fallback
        ("other/int16.npy", rng.integers(-3000, 3000, (modal_ch + 3, 1200)).astype(np.int16)),
        ("other/empty.npy", np.zeros((modal_ch, 0))),
        ("other/tiny_values.npy", rng.standard_normal((modal_ch, 128)) * 1e-7),          # fp16 subnormals after the cast
    ]
    rows = []
    for rel, arr in specs:
        if "schoffelen" in rel and modal_ch < 273:
            continue        # the reader asserts on 273 rows for a 208-channel model (covered separately)
        p = os.path.join(root, rel)
        np.save(p, arr)
        rows.append({"eeg": {"path": p}, "sentence": f"hello {os.path.basename(rel)}", "language": "English",
                     "duration": arr.shape[1] / 200})
    # a version-2 header
    p = os.path.join(root, "other", "v2.npy")
    with open(p, "wb") as f:
        np.lib.format.write_array(f, rng.standard_normal((modal_ch, 333)), version=(2, 0))
    rows.append({"eeg": {"path": p}, "sentence": "hello v2", "language": "English", "duration": 333 / 200})
    jl = os.path.join(root, f"cases_{modal_ch}.jsonl")
    with open(jl, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    return jl


def datasets(jl, proc, modal_ch):
    from utils.reader import CustomDataset
    kw = dict(data_list_path=jl, processor=proc, modal="eeg", modal_ch=modal_ch, mode="val", sample_rate=200,
              orig_sample_rate=200, language="English", timestamps=False, min_duration=0.5, max_duration=30)
    return CustomDataset(**kw), CustomDataset(raw_signals=True, **kw)
