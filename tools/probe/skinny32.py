import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from neuspeech1_amd import ops
from neuspeech1_amd.ops import rowmap
dev = torch.device("cuda:0")
M = 96000
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
for K in (512, 1536, 2048):
    A = torch.randn(M, K, device=dev).half(); B = (torch.randn(32, K, device=dev) * 0.05).half()
    C = torch.empty(M, 32, device=dev, dtype=torch.float16)
    ms = t(lambda: ops.gemm(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=32, C16=C, c16m=rowmap(32)))
    print(f"N=32 K={K}: {ms*1000:.1f} us  {M*K*2/ms/1e9:.2f} TB/s", flush=True)
