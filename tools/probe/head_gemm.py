"""LM-head GEMM of the decode step (M = 128 / 640 rows, N = 51968, K = 512) per kernel variant, graph-replay timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
L = lib.load()
N, K = 51968, 512
B = (torch.randn(N, K, device=dev) * 0.05).half()
for M in (128, 256, 640, 1024):
    A = (torch.randn(M, K, device=dev)).half()
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    row = []
    for mode in (1, 0, 2, 3, 4):
        L.ns_debug_set_ring(mode)
        row.append((mode, t(lambda: ops.gemm(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=C, c16m=rowmap(N)))))
    L.ns_debug_set_ring(1)
    byts = N * K * 2 + M * N * 2
    print(f"M={M:5d} " + "  ".join(f"{m if isinstance(m, str) else 'mode' + str(m)}: {ms*1e3:6.1f}us" for m, ms in row) + f"   HBM floor {byts/5.3e6:.1f}us", flush=True)
