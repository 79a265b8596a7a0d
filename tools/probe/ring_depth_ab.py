"""ns_gemm_ring's 128-row tile with a six-slice ring walked in pairs (96 KiB of LDS: ONE workgroup per CU) against four
single slices (64 KiB: two per CU), same process, interleaved (A/B flag 1 of ns_debug_set_ring: 101 = four slices, 100 = pairs).
Shapes: the split-K LM-head input gradient (M 2816, N 512, K 51 968, 12 splits: 1 056 workgroups) and mid-size launches of > 384 tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap
dev = torch.device("cuda:0")
L = lib.load()
F16 = torch.float16
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(F16)


def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = []
M, N, K = 2816, 512, 51968
A, B = rnd(M, K, scale=0.05), rnd(N, K, scale=0.05)
C32 = torch.zeros(M, N, device=dev)
cases.append(("LM-head dgrad split-K 12", lambda: ops.gemm(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C32=C32, ldc32=N, splits=12), 2.0 * M * N * K))
for (M2, N2, K2) in ((8192, 1024, 2048), (16384, 512, 512), (12288, 2048, 512), (6144, 1536, 1536)):
    A2, B2 = rnd(M2, K2), rnd(N2, K2, scale=0.05)
    C2 = torch.empty(M2, N2, device=dev, dtype=F16)
    L.ns_debug_set_ring(2)      # force the 128^2 ring for these
    cases.append((f"ring128 M={M2} N={N2} K={K2}", (lambda A2=A2, B2=B2, C2=C2, M2=M2, N2=N2, K2=K2: ops.gemm(A=A2, am=rowmap(K2), K=K2, B=B2, ldb=K2, M=M2, N=N2, C16=C2, c16m=rowmap(N2))), 2.0 * M2 * N2 * K2))
# the decoder-side launches of a training step (2 816 rows: the 64-row tile)
Ms = 2816
for (N3, K3) in ((512, 512), (1536, 512), (2048, 512), (512, 2048), (512, 1536), (1024, 512)):
    A3, B3 = rnd(Ms, K3), rnd(N3, K3, scale=0.05)
    C3 = torch.empty(Ms, N3, device=dev, dtype=F16)
    cases.append((f"ring64 M={Ms} N={N3} K={K3}", (lambda A3=A3, B3=B3, C3=C3, N3=N3, K3=K3: ops.gemm(A=A3, am=rowmap(K3), K=K3, B=B3, ldb=K3, M=Ms, N=N3, C16=C3, c16m=rowmap(N3))), 2.0 * Ms * N3 * K3))
for name, fn, fl in cases:
    L.ns_debug_set_ring(1 if ("split" in name or "ring64" in name) else 2)
    six, four = (1, 0) if "ring64" not in name else (0, 2)       # flag values that select the six-slice / four-slice form of this tile
    best = {six: 1e9, four: 1e9}
    for rep in range(5):
        for flag in (six, four):
            L.ns_debug_set_ring(100 + flag)
            best[flag] = min(best[flag], t(fn))
    print(f"{name:40s} pairs/6 slices {best[six]:7.1f} us ({fl / best[six] / 1e6:6.0f} TF)   4 slices {best[four]:7.1f} us ({fl / best[four] / 1e6:6.0f} TF)   4/6 = {best[four] / best[six]:.3f}", flush=True)
L.ns_debug_set_ring(100); L.ns_debug_set_ring(1)
