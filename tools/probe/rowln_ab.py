"""ns_gemm_ln against ns_gemm + ns_layernorm_fwd at the step's shapes (M = 96 000, N = 512; K = 512 with / without the LoRA second
product, K = 2048), operands rotated over NSET buffer sets so that nothing is served by the Infinity Cache from the previous round."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuspeech1_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N, NSET, REP = int(os.environ.get("M", 96000)), 512, 3, 6


def run(K, K2):
    sets = []
    for s in range(NSET):
        g = torch.Generator(device=dev).manual_seed(s)
        r = lambda *sh, sc=1.0, dt=torch.float16: (torch.randn(*sh, device=dev, generator=g) * sc).to(dt)  # noqa: E731
        d = dict(A=r(M, K), W=r(N, K, sc=K ** -0.5), bias=r(N, sc=0.1, dt=torch.float32), R=r(M, N, sc=2.0, dt=torch.float32),
                 gamma=r(N, dt=torch.float32), beta=r(N, dt=torch.float32), H=torch.empty(M, N, device=dev),
                 x=torch.empty(M, N, device=dev, dtype=torch.float16), m=torch.empty(M, device=dev), rs=torch.empty(M, device=dev),
                 u=r(M, 32, sc=0.5), sB=r(N, 32, sc=0.2))
        sets.append(d)

    def kw(d):
        k = dict(A=d["A"], am=ops.rowmap(K), K=K, B=d["W"], ldb=K, M=M, N=N, bias=d["bias"], R32=d["R"], H32=d["H"], h32m=ops.rowmap(N))
        if K2:
            k.update(A2=d["u"], am2=ops.rowmap(32), K2=K2, B2=d["sB"], ldb2=32)
        return k

    def two(d):
        ops.gemm(**kw(d))
        ops.layernorm_fwd(d["H"], d["gamma"], d["beta"], d["x"], d["m"], d["rs"], M, N)

    def one(d):
        ops.gemm_ln(gamma=d["gamma"], beta=d["beta"], x16=d["x"], ldx=N, mean=d["m"], rstd=d["rs"], **kw(d))
    out = {}
    for name, fn in (("gemm+ln", two), ("gemm_ln", one)):
        for d in sets:
            fn(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            for d in sets:
                fn(d)
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / (REP * NSET) * 1e3
    byt = M * (K * 2 + N * 4 * 2 + N * 2) / 1e6
    print(f"K={K:5d} K2={K2:2d}: gemm+ln {out['gemm+ln']:7.1f} us   gemm_ln {out['gemm_ln']:7.1f} us   "
          f"({byt:.0f} MB algorithmic = {byt / out['gemm_ln']:.2f} TB/s; {2 * M * N * K / out['gemm_ln'] / 1e6:.0f} TFLOP/s)", flush=True)


for K, K2 in ((512, 0), (512, 32), (2048, 32)):
    run(K, K2)
