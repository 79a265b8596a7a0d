"""Cost of the LoRA-dropout mask in the dgrad GEMM: same shapes with drop_p = 0.05 and 0."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops
from neuspeech1_amd.ops import rowmap, NS_GEMM_DGELU
dev = torch.device("cuda:0")
M, d, f, r = 96000, 512, 2048, 32
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).half()
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for name, K, N, K2, dg in (("dx<-qkv", 3 * d, d, 3 * r, False), ("dattn<-out", d, d, r, False), ("dx2<-fc1", f, d, r, False), ("dgelu<-fc2", d, f, r, True)):
    A, B = rnd(M, K), rnd(N, K, scale=.04)
    A2, B2 = rnd(M, K2), rnd(N, K2, scale=.1)
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    P = rnd(M, N)
    kw = dict(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=C, c16m=rowmap(N), A2=A2, am2=rowmap(K2), K2=K2, B2=B2, ldb2=K2)
    if dg: kw.update(P16=P, p16m=rowmap(N), flags=NS_GEMM_DGELU)
    a = t(lambda: ops.gemm(drop_p=0.05, drop_seed=7, **kw))
    b = t(lambda: ops.gemm(**kw))
    print(f"{name:12s} drop {a*1000:7.1f} us   no-drop {b*1000:7.1f} us   (+{(a-b)*1000:.1f} us)", flush=True)
