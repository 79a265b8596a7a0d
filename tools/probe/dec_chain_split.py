"""Would the training step's decoder-side chain (small launches at M = B x L = 2816 rows: 10-28 us each, far from filling the GPU) gain from
running as two / four concurrent half-batch chains (graph branches)?  One decoder layer's six GEMMs (q|k|v, out, cross-q, cross-out, fc1 + GELU,
fc2 + residual) + three LayerNorms, x 6 layers, captured: one chain at M rows against `nsplit` chains at M / nsplit rows on separate streams."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuspeech1_amd import ops  # noqa: E402
from neuspeech1_amd.ops import NS_GEMM_GELU, rowmap  # noqa: E402

dev = torch.device("cuda:0")
M, d, f = 2816, 512, 2048
r = lambda *s, dt=torch.float16, sc=1.0: (torch.randn(*s, device=dev) * sc).to(dt)  # noqa: E731
W = dict(qkv=r(3 * d, d, sc=d ** -0.5), out=r(d, d, sc=d ** -0.5), cq=r(d, d, sc=d ** -0.5), cout=r(d, d, sc=d ** -0.5),
         fc1=r(f, d, sc=d ** -0.5), fc2=r(d, f, sc=f ** -0.5))
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
h = [torch.randn(M, d, device=dev) for _ in range(2)]
x16, qkv, q, gf = r(M, d), r(M, 3 * d), r(M, d), r(M, f)
st = (torch.empty(M, device=dev), torch.empty(M, device=dev))


def lin(x, w, rows, N, K, **kw):
    ops.gemm(A=x, am=rowmap(K), K=K, B=w, ldb=K, M=rows, N=N, **kw)


def chain(r0, r1):
    n = r1 - r0
    ha, hb, x_, qkv_, q_, gf_ = h[0][r0:r1], h[1][r0:r1], x16[r0:r1], qkv[r0:r1], q[r0:r1], gf[r0:r1]
    s_ = (st[0][r0:r1], st[1][r0:r1])
    for _ in range(6):
        ops.layernorm_fwd(ha, gamma, beta, x_, *s_, n, d)
        lin(x_, W["qkv"], n, 3 * d, d, C16=qkv_, c16m=rowmap(3 * d))
        lin(x_, W["out"], n, d, d, R32=ha, H32=hb, h32m=rowmap(d))
        ops.layernorm_fwd(hb, gamma, beta, x_, *s_, n, d)
        lin(x_, W["cq"], n, d, d, C16=q_, c16m=rowmap(d))
        lin(q_, W["cout"], n, d, d, R32=hb, H32=ha, h32m=rowmap(d))
        ops.layernorm_fwd(ha, gamma, beta, x_, *s_, n, d)
        lin(x_, W["fc1"], n, f, d, G16=gf_, g16m=rowmap(f), flags=NS_GEMM_GELU)
        lin(gf_, W["fc2"], n, d, f, R32=ha, H32=hb, h32m=rowmap(d))
        ha, hb = hb, ha


def run(nsplit):
    cuts = [M * k // nsplit // 64 * 64 for k in range(nsplit)] + [M]
    side = [torch.cuda.Stream(dev) for _ in range(nsplit - 1)]

    def body():
        main = torch.cuda.current_stream()
        for k, sd in enumerate(side):
            sd.wait_stream(main)
            with torch.cuda.stream(sd):
                chain(cuts[k + 1], cuts[k + 2])
        chain(cuts[0], cuts[1])
        for sd in side:
            main.wait_stream(sd)
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


for n in (1, 2, 4, 1, 2, 4):
    print(f"{n} chain(s): {run(n):.3f} ms per 6-layer pass (54 launches per chain)", flush=True)
