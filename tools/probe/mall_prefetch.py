"""Does a read of the NEXT layer's cross K|V during the decode step's latency-bound chain make its cross-attention faster?  The greedy
cross-attention (ns_attn_decode, 393 MB of K|V per layer at B = 128, six layers rotated = 2.36 GB per step: HBM-cold) timed alone, and
right behind a kernel that has just read the whole buffer (Infinity Cache = 256 MB memory-side) or its first `FRAC` part."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuspeech1_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib.load()
B, S, H, d, NL, REP = 128, 1500, 8, 64, 6, 5
D = H * d
g = torch.Generator(device=dev).manual_seed(0)
kv = [(torch.randn(B * S, 2 * D, device=dev, generator=g) * 0.5).half() for _ in range(NL)]
q = (torch.randn(B, D, device=dev, generator=g) * 0.5).half()
o = torch.empty(B, D, device=dev, dtype=torch.float16)


def attn(i):
    ops.attn_decode(Q=q, K=kv[i], V=(kv[i], D), O=o, groups=B, nq=1, H=H, Lk=S, Lk_max=S, ldq=D, ldk=2 * D, ldv=2 * D, ldo=D, kv_group_stride=S)


def timed(pre):
    tot = 0.0
    for _ in range(REP):
        for i in range(NL):
            if pre is not None:
                pre(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            attn(i)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
    return tot * 1e3 / (REP * NL)


for i in range(NL):
    attn(i)
torch.cuda.synchronize()
print(f"cross-attention alone (cold, rotated over {NL} layers): {timed(None):6.1f} us", flush=True)
for frac in (1.0, 0.6, 0.4, 0.25):
    rows = int(B * S * frac)
    print(f"behind a read of the first {frac:4.2f} of its K|V ({rows * 2 * D * 2 / 1e6:5.0f} MB): {timed(lambda i: kv[i][:rows].float().sum()):6.1f} us", flush=True)
print(f"same buffer twice in a row (second launch): {timed(lambda i: attn(i)):6.1f} us", flush=True)
