"""Encoder attention backward at the bench shape (B 64, H 8, S 1500, random data): two-pass kernels against the one-pass
kernel (csrc/ns_attn_bwd1.hip), interleaved rounds in one process (cdna guide rule 24)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from neuspeech1_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, S = int(os.environ.get("B", 64)), 8, 1500
LQ = int(os.environ.get("LQ", S))     # LQ=45: the decoder's cross-attention shape
d = H * 64
g = torch.Generator(device=dev).manual_seed(1)
qkv = (torch.randn(B * S, 3 * d, device=dev, generator=g) * 0.5).half()
qx = (torch.randn(B * LQ, d, device=dev, generator=g) * 0.5).half()
O = torch.zeros(B * LQ, d, device=dev, dtype=torch.float16)
LSE = torch.zeros(B, H, LQ, device=dev)
common = dict(Q=qx if LQ != S else qkv, K=(qkv, d), V=(qkv, 2 * d), O=O, B=B, H=H, Lq=LQ, Lk=S, ldq=d if LQ != S else 3 * d, ldk=3 * d, ldv=3 * d, ldo=d, causal=False, LSE=LSE)
ops.attn_fwd(**common)
dO = (torch.randn(B * LQ, d, device=dev, generator=g) * 0.5).half()
dqx = torch.zeros(B * LQ, d, device=dev, dtype=torch.float16)
dqkv = torch.zeros(B * S, 3 * d, device=dev, dtype=torch.float16)
Delta = torch.zeros(B, H, LQ, device=dev)
NEED = ops.attn_bwd_workspace_bytes(B, H, LQ, S)
ws = torch.zeros(NEED + B * H * 8 * 32, device=dev, dtype=torch.uint8)   # + room for the diagnostic build's stamps
bw = dict(dO=dO, dQ=dqx if LQ != S else dqkv, dK=(dqkv, d), dV=(dqkv, 2 * d), Delta=Delta, lddo=d, lddq=d if LQ != S else 3 * d, lddk=3 * d, lddv=3 * d)


def t(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {"fwd": [], "two": [], "one": []}
for _ in range(4):
    res["fwd"].append(t(lambda: ops.attn_fwd(**common)))
    res["two"].append(t(lambda: ops.attn_bwd(**common, **bw)))
    res["one"].append(t(lambda: ops.attn_bwd(**common, **bw, workspace=ws)))
fl = 4.0 * B * H * LQ * S * 64
for k, v in res.items():
    ms = min(v)
    n = 1 if k == "fwd" else 2.5
    print(f"{k}: min {ms:.3f} ms  median {sorted(v)[len(v) // 2]:.3f} ms  -> {n * fl / ms / 1e9:.0f} TFLOP/s algorithmic")

if os.environ.get("NS_EXTRA_HIPCC_FLAGS", "").find("NS_AB1_STAMPS") >= 0:
    torch.cuda.synchronize()
    st = ws[NEED:].view(torch.int64).view(B * H, 8, 4).double()
    tot = st[:, :, 3].mean().item()
    print(f"stamps (cycles per workgroup-wave, mean): total {tot:.0f}; barrier {st[:, :, 0].mean().item():.0f} ({100 * st[:, :, 0].mean().item() / tot:.1f} %), "
          f"first half {st[:, :, 1].mean().item():.0f}, second half {st[:, :, 2].mean().item():.0f}")
    print("per wave barrier share:", [round(st[:, w, 0].mean().item() / tot, 3) for w in range(8)])
    print("per wave half1:", [round(st[:, w, 1].mean().item() / tot, 3) for w in range(8)], "half2:", [round(st[:, w, 2].mean().item() / tot, 3) for w in range(8)])
