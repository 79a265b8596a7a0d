"""Race screen for the LDS-DMA GEMM kernels: the NT kernels are deterministic (no atomics), so repeated launches on
the same inputs must agree bit for bit, while other work (a second stream hammering HBM) perturbs the timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap, NS_GEMM_GELU
dev = torch.device("cuda:0")
torch.manual_seed(0)
REP = int(os.environ.get("REP", 150))
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).half()
side = torch.cuda.Stream()
junk = torch.empty(64 * 1024 * 1024, device=dev)
cases = [("qkv+lora", 96000, 1536, 512, 96, 512, 0.0), ("fc2", 96000, 512, 2048, 0, 0, 0.0), ("ragged K=528 M=95999+", 95872 + 77, 512, 528, 32, 0, 0.0),
         ("dgrad+lora+drop", 96000, 512, 2048, 32, 0, 0.05), ("conv-like K=1536", 192000, 512, 1536, 0, 0, 0.0)]
L = lib.load()
for mode in (4, 3, 2):
    L.ns_debug_set_ring(mode)
    for name, M, N, K, K2, ng, dp in cases:
        A, B = rnd(M, K), rnd(N, K, sc=0.05)
        kw = dict(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=torch.randn(N, device=dev))
        if K2:
            kw.update(A2=rnd(M, K2), am2=rowmap(K2), K2=K2 // (3 if ng else 1), B2=rnd(N, K2 // (3 if ng else 1), sc=0.1), ldb2=K2 // (3 if ng else 1), a2_ngroup=ng)
        if dp:
            kw.update(drop_p=dp, drop_seed=77)
        C0 = torch.empty(M, N, device=dev, dtype=torch.float16)
        ops.gemm(C16=C0, c16m=rowmap(N), **kw)
        torch.cuda.synchronize()
        bad = 0
        C = torch.empty_like(C0)
        for i in range(REP):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    junk.add_(1.0)          # timing perturbation from another stream
            C.fill_(float("nan"))
            ops.gemm(C16=C, c16m=rowmap(N), **kw)
            if not torch.equal(C, C0):
                bad += 1
        torch.cuda.synchronize()
        print(f"mode {mode} {name:24s} {REP} launches, mismatching: {bad}", flush=True)
L.ns_debug_set_ring(1)
