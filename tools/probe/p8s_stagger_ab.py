"""Persistent GEMM with a start-up stagger of its workgroups: group g = (workgroup >> 3) & 3 starts g * S / 4 cycles late.
profiles/r6_probe_p8s_stagger.log is the SWEEP of S over every epilogue kind, taken with a probe build that applied the stagger to all kinds and
read S from the A/B flag (ns_debug_set_ring(100 + S)); only the gelu'-multiply kind gained, and the shipped kernel staggers that kind alone by
S = 24 000 cycles (csrc/ns_gemm_p8s.hip).  Against the shipped library this script therefore compares stagger on (flag 0) with stagger off
(flag 8) per case: SW = [0, 8]; the other kinds print equal times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap, NS_GEMM_GELU, NS_GEMM_GELU_SAVE_GRAD
dev = torch.device("cuda:0")
L = lib.load()
M, d, f, r = 96000, 512, 2048, 32
F16, F32 = torch.float16, torch.float32
rnd = lambda *s, dtype=F16, scale=1.0: (torch.randn(*s, device=dev) * scale).to(dtype)
x, xf = rnd(M, d), rnd(M, f)
Wqkv, W1, W2t, Wo, Wkv = rnd(3 * d, d, scale=.04), rnd(f, d, scale=.04), rnd(d, f, scale=.04), rnd(d, d, scale=.04), rnd(12 * d, d, scale=.04)
b3, b1 = rnd(3 * d, dtype=F32), rnd(f, dtype=F32)
o3, of_, og, od, okv = (torch.empty(M, n, device=dev, dtype=F16) for n in (3 * d, f, f, d, 12 * d))
P = rnd(M, f)
R, H = rnd(M, d, dtype=F32), torch.empty(M, d, device=dev, dtype=F32)
cases = {
 "q|k|v (plain, N 1536)": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wqkv, ldb=d, M=M, N=3*d, bias=b3, C16=o3, c16m=rowmap(3*d)), 2.*M*3*d*d),
 "stacked cross K|V (plain, N 6144)": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wkv, ldb=d, M=M, N=12*d, C16=okv, c16m=rowmap(12*d)), 2.*M*12*d*d),
 "fc1 gelu + saved gelu' (N 2048)": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, bias=b1, C16=of_, c16m=rowmap(f), G16=og, g16m=rowmap(f), flags=NS_GEMM_GELU | NS_GEMM_GELU_SAVE_GRAD), 2.*M*f*d),
 "fc2 dgrad x gelu' (N 2048)": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, C16=of_, c16m=rowmap(f), P16=P, p16m=rowmap(f), flags=ops.NS_GEMM_MUL_P16), 2.*M*f*d),
 "out_proj dgrad (K 512, N 512)": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wo, ldb=d, M=M, N=d, C16=od, c16m=rowmap(d)), 2.*M*d*d),
 "fc2 + residual (K 2048, N 512)": (lambda: ops.gemm(A=xf, am=rowmap(f), K=f, B=W2t, ldb=f, M=M, N=d, R32=R, H32=H, h32m=rowmap(d)), 2.*M*d*f),
}
SW = [0, 8]


def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, (fn, flops) in cases.items():
    best = {s: 1e9 for s in SW}
    fn(); fn(); torch.cuda.synchronize()
    for rep in range(5):
        for s in SW:
            L.ns_debug_set_ring(100 + s)
            best[s] = min(best[s], t(fn))
    print(f"{name:36s} " + "  ".join(f"{'stagger on ' if s == 0 else 'stagger off'}: {best[s]:7.1f} us" for s in SW), flush=True)
L.ns_debug_set_ring(100)
