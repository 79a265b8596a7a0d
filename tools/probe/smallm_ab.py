"""Decode-shaped GEMMs (M = 128 / 640): small-M split-K kernel (mode 1) against the 128x32 register-staged tile
(mode 6), time per launch and max |difference| of the outputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap, NS_GEMM_GELU
dev = torch.device("cuda:0")
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).half()
def t(fn, n=50):
    """time per launch from a hipGraph replay of n launches (eager launches from Python cost ~7 us each on the host)"""
    fn(); fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
L = lib.load()
for M in (128, 640, 8, 1000):
    for N, K, epi in ((1536, 512, "c16"), (512, 512, "res"), (2048, 512, "gelu"), (512, 2048, "res"), (1280, 1280, "res"), (520, 272, "c16")):
        A, B = rnd(M, K), rnd(N, K)
        bias = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev)
        outs = []
        row = []
        for mode in (6, 1):
            L.ns_debug_set_ring(mode)
            C = torch.zeros(M, N, device=dev, dtype=torch.float16); G = torch.zeros_like(C); H = torch.zeros_like(R)
            kw = dict(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias)
            if epi == "c16": kw.update(C16=C, c16m=rowmap(N))
            elif epi == "res": kw.update(R32=R, H32=H, h32m=rowmap(N))
            else: kw.update(C16=C, c16m=rowmap(N), G16=G, g16m=rowmap(N), flags=NS_GEMM_GELU)
            row.append(t(lambda: ops.gemm(**kw)))
            outs.append((C.float().clone(), G.float().clone(), H.clone()))
        L.ns_debug_set_ring(1)
        diff = max((a - b).abs().max().item() for a, b in zip(*outs))
        ref = (A.float() @ B.float().T + bias)
        err = (outs[1][2] - R - ref).abs().max().item() if epi == "res" else (outs[1][0] - ref).abs().max().item()
        print(f"M={M:4d} N={N:5d} K={K:5d} {epi:5s} old {row[0]*1000:6.1f}us  new {row[1]*1000:6.1f}us   max|old-new| {diff:.2e}  |new-ref| {err:.2e}", flush=True)
