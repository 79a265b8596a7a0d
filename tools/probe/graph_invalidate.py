"""What happens to engine.train_step's hipGraph capture when ANOTHER host thread, outside the package's capture lock,
synchronizes the device and allocates pinned / device memory meanwhile (torch's pin-memory thread does the latter)?
Prints whether the capture survived, whether the eager fallback took over, and that training went on either way.
  python tools/probe/graph_invalidate.py [sync|pin|malloc|copy]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg  # noqa: E402
from neuspeech1_amd.weights import TINY, make_lora_state, make_state_dict, synth_batch  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "pin"
dev = torch.device("cuda:0")
dims = TINY
eng = MegWhisperEngine(dims, make_state_dict(dims, 42), lora=LoraSpec(r=32, alpha=64.0, dropout=0.05),
                       lora_sd=make_lora_state(dims, 32), train_cfg=TrainCfg(lr=1e-3), device=dev)
x, labels = synth_batch(dims, 3, 77)
xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
stop = False


def disturb():
    torch.cuda.set_device(dev)
    s = torch.cuda.Stream(dev)
    n = 0
    while not stop:
        n += 1
        if what == "sync":
            torch.cuda.synchronize(dev)
        elif what == "pin":
            t = torch.empty(1 << 16 + (n % 7), dtype=torch.uint8, pin_memory=True)
            del t
        elif what == "malloc":
            t = torch.empty((1 << 20) + 4096 * (n % 13), dtype=torch.uint8, device=dev)
            del t
            torch.cuda.empty_cache()
        elif what == "copy":
            with torch.cuda.stream(s):
                a = torch.empty(1 << 20, dtype=torch.uint8, pin_memory=True)
                a.to(dev, non_blocking=True)
        time.sleep(0.0005)


th = threading.Thread(target=disturb, daemon=True)
th.start()
losses = []
for i in range(8):
    losses.append(eng.train_step(xd, ld).item())
stop = True
th.join()
print(f"[{what}] graphs captured: {len(eng._graphs)}, capture failures: {eng._graph_failures}, use_graph: {eng.use_graph}, "
      f"losses {losses[0]:.4f} -> {losses[-1]:.4f}, steps {eng.step_dev.item()}")
