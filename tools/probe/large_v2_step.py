"""BASELINE configs[4] alone (bench.large_v2_leg): for `rocprofv3 --kernel-trace --stats` with NS_TRAIN_GRAPH=0 (eager launches)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

print(json.dumps(bench.large_v2_leg(torch.device("cuda:0"), B=int(os.environ.get("B", 64)), steps=int(os.environ.get("STEPS", 2)))))
