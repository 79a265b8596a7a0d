"""Decoder-side GEMMs of the training step (M = 64 x 44 = 2816 rows) per kernel variant."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap, NS_GEMM_GELU
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 2816))
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).half()
def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
L = lib.load()
for N, K, epi in ((1536, 512, "c16"), (512, 512, "res"), (2048, 512, "gelu"), (512, 2048, "res"), (512, 1024, "res")):
    A, B = rnd(M, K), rnd(N, K)
    C = torch.empty(M, N, device=dev, dtype=torch.float16); G = torch.empty_like(C)
    R = torch.randn(M, N, device=dev); H = torch.empty_like(R); bias = torch.randn(N, device=dev)
    kw = dict(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias)
    if epi == "c16": kw.update(C16=C, c16m=rowmap(N))
    elif epi == "res": kw.update(R32=R, H32=H, h32m=rowmap(N))
    else: kw.update(C16=C, c16m=rowmap(N), G16=G, g16m=rowmap(N), flags=NS_GEMM_GELU)
    row = []
    for mode in (0, 2, 4):
        L.ns_debug_set_ring(mode)
        row.append((mode, t(lambda: ops.gemm(**kw))))
    L.ns_debug_set_ring(1)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N:5d} K={K:5d} {epi:5s} " + "  ".join(f"mode{m}: {ms*1000:6.1f}us {fl/ms/1e9:5.0f}TF" for m, ms in row), flush=True)
