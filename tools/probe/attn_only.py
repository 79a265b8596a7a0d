"""Encoder attention forward / one-pass backward alone at the bench shape, for counter passes (rocprofv3 --pmc ...): WHAT=fwd|bwd, N launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from neuspeech1_amd import ops
dev = torch.device("cuda:0")
B, H, S = int(os.environ.get("B", 64)), int(os.environ.get("H", 8)), 1500
d = H * 64
g = torch.Generator(device=dev).manual_seed(1)
qkv = (torch.randn(B * S, 3 * d, device=dev, generator=g) * 0.5).half()
O = torch.zeros(B * S, d, device=dev, dtype=torch.float16)
LSE = torch.zeros(B, H, S, device=dev)
common = dict(Q=qkv, K=(qkv, d), V=(qkv, 2 * d), O=O, B=B, H=H, Lq=S, Lk=S, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, causal=False, LSE=LSE)
ops.attn_fwd(**common)
dO = (torch.randn(B * S, d, device=dev, generator=g) * 0.5).half()
dqkv = torch.zeros(B * S, 3 * d, device=dev, dtype=torch.float16)
Delta = torch.zeros(B, H, S, device=dev)
ws = torch.zeros(ops.attn_bwd_workspace_bytes(B, H, S, S), device=dev, dtype=torch.uint8)
bw = dict(dO=dO, dQ=dqkv, dK=(dqkv, d), dV=(dqkv, 2 * d), Delta=Delta, lddo=d, lddq=3 * d, lddk=3 * d, lddv=3 * d)
what, n = os.environ.get("WHAT", "fwd"), int(os.environ.get("N", 5))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    if what == "fwd":
        ops.attn_fwd(**common)
    else:
        ops.attn_bwd(**common, **bw, workspace=ws)
e1.record()
torch.cuda.synchronize()
print(what, "ms per launch", e0.elapsed_time(e1) / n)
