"""Greedy cross-attention (ns_attn_decode, B = 128, 8 heads, 1500 keys) on the K|V layout the projection writes -- rows of all heads,
a head's 128-B segment every 2 KB -- against a HEAD-MAJOR image (each (sequence, head) a contiguous 192 KB block of K and of V), emulated
with groups = B * H, H = 1.  Six layers rotated (HBM-cold), us per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuspeech1_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib.load()
B, S, H, d, NL, REP = 128, 1500, 8, 64, 6, 10
D = H * d
g = torch.Generator(device=dev).manual_seed(0)
kv = [(torch.randn(B * S, 2 * D, device=dev, generator=g) * 0.5).half() for _ in range(NL)]
q = (torch.randn(B, D, device=dev, generator=g) * 0.5).half()
o = torch.empty(B, D, device=dev, dtype=torch.float16)
# head-major copies: K' / V' [B, H, S, 64]
kh = [t.view(B, S, 2, H, d)[:, :, 0].permute(0, 2, 1, 3).contiguous() for t in kv]
vh = [t.view(B, S, 2, H, d)[:, :, 1].permute(0, 2, 1, 3).contiguous() for t in kv]
o2 = torch.empty(B * H, d, device=dev, dtype=torch.float16)


def rows(i):
    ops.attn_decode(Q=q, K=kv[i], V=(kv[i], D), O=o, groups=B, nq=1, H=H, Lk=S, Lk_max=S, ldq=D, ldk=2 * D, ldv=2 * D, ldo=D, kv_group_stride=S)


def heads(i):
    ops.attn_decode(Q=q.view(B * H, d), K=kh[i].view(B * H * S, d), V=vh[i].view(B * H * S, d), O=o2, groups=B * H, nq=1, H=1, Lk=S, Lk_max=S,
                    ldq=d, ldk=d, ldv=d, ldo=d, kv_group_stride=S)


def timed(fn):
    lst = ops.LaunchList()
    with ops.recording(lst):
        for i in range(NL):
            fn(i)
    lst.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        lst.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (REP * NL)


rows(0); heads(0)
torch.cuda.synchronize()
print("max |diff|", (o.view(B * H, d).float() - o2.float()).abs().max().item())
print(f"row layout (as projected): {timed(rows):6.1f} us    head-major: {timed(heads):6.1f} us   ({B * S * 2 * D * 2 / 1e6:.0f} MB)", flush=True)
