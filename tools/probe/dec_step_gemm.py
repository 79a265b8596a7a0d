"""The beam-5 decode step's GEMM shapes (M = 640 rows) per kernel variant (ns_debug_set_ring modes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap, NS_GEMM_GELU
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 640))
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).half()
def t(fn, n=30):
    """us-scale launches: replayed from a launch list (the clock sees the kernels, not ctypes)"""
    fn(); fn(); torch.cuda.synchronize()
    lst = ops.LaunchList()
    with ops.recording(lst):
        for _ in range(n): fn()
    lst.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lst.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
L = lib.load()
modes = [int(m) for m in os.environ.get("MODES", "1,0,2,4").split(",")]
for N, K, epi in ((1536, 512, "c16"), (2048, 512, "gelu"), (512, 512, "res"), (512, 2048, "res"), (51968, 512, "c16")):
    A, B = rnd(M, K), rnd(N, K)
    C = torch.empty(M, N, device=dev, dtype=torch.float16); G = torch.empty_like(C) if N < 10000 else None
    R = torch.randn(M, N, device=dev) if epi == "res" else None
    H = torch.empty_like(R) if epi == "res" else None
    bias = torch.randn(N, device=dev)
    kw = dict(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias)
    if epi == "c16": kw.update(C16=C, c16m=rowmap(N))
    elif epi == "res": kw.update(R32=R, H32=H, h32m=rowmap(N))
    else: kw.update(G16=G, g16m=rowmap(N), flags=NS_GEMM_GELU)
    row = []
    for mode in modes:
        L.ns_debug_set_ring(mode)
        try:
            row.append((mode, t(lambda: ops.gemm(**kw))))
        except Exception as e:
            row.append((mode, float("nan")))
    L.ns_debug_set_ring(1)
    print(f"M={M} N={N:5d} K={K:5d} {epi:5s} " + "  ".join(f"mode{m}: {ms*1000:6.1f}us" for m, ms in row), flush=True)
