"""Same-process A/B of GEMM kernel variants (ns_debug_set_ring modes) on the step's shapes; alternates modes per repeat."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap, NS_GEMM_GELU, NS_GEMM_DGELU
dev = torch.device("cuda:0")
modes = [int(m) for m in os.environ.get("MODES", "3,4").split(",")]
M, d, f, r = 96000, 512, 2048, 32
F16, F32 = torch.float16, torch.float32
rnd = lambda *s, dtype=F16, scale=1.0: (torch.randn(*s, device=dev) * scale).to(dtype)
x, xf = rnd(M, d), rnd(M, f)
Wqkv, W1, W2, Wo = rnd(3 * d, d, scale=.04), rnd(f, d, scale=.04), rnd(d, f, scale=.04), rnd(d, d, scale=.04)
b3, b1, bd = rnd(3 * d, dtype=F32), rnd(f, dtype=F32), rnd(d, dtype=F32)
u3, B3 = rnd(M, 3 * r), rnd(3 * d, r, scale=.1)
u1, Bl = rnd(M, r), rnd(d, r, scale=.1)
o3, of_, og, od = torch.empty(M, 3 * d, device=dev, dtype=F16), torch.empty(M, f, device=dev, dtype=F16), torch.empty(M, f, device=dev, dtype=F16), torch.empty(M, d, device=dev, dtype=F16)
h, ho = rnd(M, d, dtype=F32), torch.empty(M, d, device=dev, dtype=F32)
cases = {
 "qkv c16": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wqkv, ldb=d, M=M, N=3*d, bias=b3, C16=o3, c16m=rowmap(3*d)), 2.*M*3*d*d),
 "qkv+lora": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wqkv, ldb=d, M=M, N=3*d, bias=b3, C16=o3, c16m=rowmap(3*d), A2=u3, am2=rowmap(3*r), K2=r, B2=B3, ldb2=r, a2_ngroup=d), 2.*M*3*d*(d+r)),
 "out+res": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wo, ldb=d, M=M, N=d, bias=bd, R32=h, H32=ho, h32m=rowmap(d)), 2.*M*d*d),
 "fc1+gelu": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, bias=b1, C16=of_, c16m=rowmap(f), G16=og, g16m=rowmap(f), flags=NS_GEMM_GELU), 2.*M*f*d),
 "fc2+res": (lambda: ops.gemm(A=xf, am=rowmap(f), K=f, B=W2, ldb=f, M=M, N=d, bias=bd, R32=h, H32=ho, h32m=rowmap(d)), 2.*M*f*d),
 "dgelu (dfc2)": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1.t().contiguous() if False else W1, ldb=d, M=M, N=f, C16=of_, c16m=rowmap(f), P16=og, p16m=rowmap(f), flags=NS_GEMM_DGELU), 2.*M*f*d),
 "dgrad+lora+drop": (lambda: ops.gemm(A=xf, am=rowmap(f), K=f, B=W2, ldb=f, M=M, N=d, C16=od, c16m=rowmap(d), A2=u1, am2=rowmap(r), K2=r, B2=Bl, ldb2=r, drop_p=0.05, drop_seed=123), 2.*M*d*(f+r)),
 "dgrad qkv+lora+drop": (lambda: ops.gemm(A=o3, am=rowmap(3*d), K=3*d, B=rnd(d, 3*d, scale=.04), ldb=3*d, M=M, N=d, C16=od, c16m=rowmap(d), A2=u3, am2=rowmap(3*r), K2=3*r, B2=rnd(d, 3*r, scale=.1), ldb2=3*r, drop_p=0.05, drop_seed=123), 2.*M*d*(3*d+3*r)),
}
L = lib.load()
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, (fn, flops) in cases.items():
    best = {m: 1e9 for m in modes}
    for rep in range(4):
        for m in modes:
            L.ns_debug_set_ring(m)
            if rep == 0:
                fn(); fn(); torch.cuda.synchronize()
            best[m] = min(best[m], t(fn))
    print(f"{name:22s} " + "  ".join(f"mode{m}: {best[m]*1000:7.1f} us {flops/best[m]/1e9:6.0f} TF/s" for m in modes), flush=True)
