"""EXPERIMENT: do the HBM-bound N = 512 launches of the one-tile kernel gain from a staggered first round (NS_P8_STAGGER cycles)?
Run once per setting: the launcher reads the variable once."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops
from neuspeech1_amd.ops import rowmap
dev = torch.device("cuda:0")
M, d, f = 96000, 512, 2048
rnd = lambda *s, dtype=torch.float16, scale=1.0: (torch.randn(*s, device=dev) * scale).to(dtype)
x, xf = rnd(M, d), rnd(M, f)
Wo, W2, Wk = rnd(d, d, scale=.04), rnd(d, f, scale=.04), rnd(d, 2 * d, scale=.04)
bd = rnd(d, dtype=torch.float32)
h, ho = rnd(M, d, dtype=torch.float32), torch.empty(M, d, device=dev)
xk = rnd(M, 2 * d)
od = torch.empty(M, d, device=dev, dtype=torch.float16)
cases = {
 "out+res  (K 512)": lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wo, ldb=d, M=M, N=d, bias=bd, R32=h, H32=ho, h32m=rowmap(d)),
 "ckv dgrad+res (K 1024)": lambda: ops.gemm(A=xk, am=rowmap(2 * d), K=2 * d, B=Wk, ldb=2 * d, M=M, N=d, R32=h, H32=ho, h32m=rowmap(d)),
 "fc2+res  (K 2048)": lambda: ops.gemm(A=xf, am=rowmap(f), K=f, B=W2, ldb=f, M=M, N=d, bias=bd, R32=h, H32=ho, h32m=rowmap(d)),
 "dgrad c16 (K 2048)": lambda: ops.gemm(A=xf, am=rowmap(f), K=f, B=W2, ldb=f, M=M, N=d, C16=od, c16m=rowmap(d)),
 "dgrad c16 (K 512)": lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wo, ldb=d, M=M, N=d, C16=od, c16m=rowmap(d)),
}
def t(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3
print("NS_P8_STAGGER =", os.environ.get("NS_P8_STAGGER", "0"), "  ".join(f"{k}: {t(fn):.1f} us" for k, fn in cases.items()))
