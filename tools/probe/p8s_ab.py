"""Persistent phase-interleaved GEMM (mode 9) against the one-tile-per-workgroup kernel (mode 4): outputs bit-compared, then timed
(same process, modes alternated per repeat).  Shapes = the training step's large-M GEMMs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap, NS_GEMM_GELU, NS_GEMM_DGELU
dev = torch.device("cuda:0")
L = lib.load()
M = int(os.environ.get("M", 96000)); d, f, r = 512, 2048, 32
F16, F32 = torch.float16, torch.float32
torch.manual_seed(0)
rnd = lambda *s, dtype=F16, scale=1.0: (torch.randn(*s, device=dev) * scale).to(dtype)
x, xf = rnd(M, d), rnd(M, f)
Wqkv, W1, W2, Wo = rnd(3 * d, d, scale=.04), rnd(f, d, scale=.04), rnd(d, f, scale=.04), rnd(d, d, scale=.04)
b3, b1, bd = rnd(3 * d, dtype=F32), rnd(f, dtype=F32), rnd(d, dtype=F32)
u3, B3 = rnd(M, 3 * r), rnd(3 * d, r, scale=.1)
u1, Bl, Bf = rnd(M, r), rnd(d, r, scale=.1), rnd(f, r, scale=.1)
Wd3, Bd3 = rnd(d, 3 * d, scale=.04), rnd(d, 3 * r, scale=.1)
sideB = rnd(32, f, scale=.05)
h = rnd(M, d, dtype=F32)
def outs():
    return dict(o3=torch.zeros(M, 3 * d, device=dev, dtype=F16), of_=torch.zeros(M, f, device=dev, dtype=F16), og=torch.zeros(M, f, device=dev, dtype=F16),
                od=torch.zeros(M, d, device=dev, dtype=F16), ho=torch.zeros(M, d, device=dev, dtype=F32), slab=torch.zeros(f // 256, M, 32, device=dev, dtype=F32))
def cases(o):
    return {
     "qkv c16": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wqkv, ldb=d, M=M, N=3*d, bias=b3, C16=o["o3"], c16m=rowmap(3*d)), 2.*M*3*d*d, ["o3"]),
     "qkv+lora+drop": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wqkv, ldb=d, M=M, N=3*d, bias=b3, C16=o["o3"], c16m=rowmap(3*d), A2=u3, am2=rowmap(3*r), K2=r, B2=B3, ldb2=r, a2_ngroup=d, drop_p=0.05, drop_seed=7), 2.*M*3*d*(d+r), ["o3"]),
     "out+res+lora": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wo, ldb=d, M=M, N=d, bias=bd, R32=h, H32=o["ho"], h32m=rowmap(d), A2=u1, am2=rowmap(r), K2=r, B2=Bl, ldb2=r), 2.*M*d*(d+r), ["ho"]),
     "fc1+gelu+save+lora": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, bias=b1, C16=o["of_"], c16m=rowmap(f), G16=o["og"], g16m=rowmap(f), flags=NS_GEMM_GELU | ops.NS_GEMM_GELU_SAVE_GRAD, A2=u1, am2=rowmap(r), K2=r, B2=Bf, ldb2=r), 2.*M*f*(d+r), ["of_", "og"]),
     "fc1+gelu+side": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, bias=b1, C16=o["of_"], c16m=rowmap(f), G16=o["og"], g16m=rowmap(f), flags=NS_GEMM_GELU | ops.NS_GEMM_GELU_SAVE_GRAD, side_B=sideB, side_ldb=f, side_n=32, side_out=o["slab"], side_drop_p=0.05, side_drop_seed=11), 2.*M*f*d, ["of_", "og", "slab"]),
     "fc2+res": (lambda: ops.gemm(A=xf, am=rowmap(f), K=f, B=W2, ldb=f, M=M, N=d, bias=bd, R32=h, H32=o["ho"], h32m=rowmap(d)), 2.*M*f*d, ["ho"]),
     "dfc2 (mul p16)": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, C16=o["of_"], c16m=rowmap(f), P16=xf, p16m=rowmap(f), flags=ops.NS_GEMM_MUL_P16), 2.*M*f*d, ["of_"]),
     "dgelu": (lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, C16=o["of_"], c16m=rowmap(f), P16=xf, p16m=rowmap(f), flags=NS_GEMM_DGELU), 2.*M*f*d, ["of_"]),
     "dgrad fc1+lora": (lambda: ops.gemm(A=xf, am=rowmap(f), K=f, B=W2, ldb=f, M=M, N=d, C16=o["od"], c16m=rowmap(d), A2=u1, am2=rowmap(r), K2=r, B2=Bl, ldb2=r), 2.*M*d*(f+r), ["od"]),
     "dgrad qkv+lora": (lambda: ops.gemm(A=o["o3"], am=rowmap(3*d), K=3*d, B=Wd3, ldb=3*d, M=M, N=d, C16=o["od"], c16m=rowmap(d), A2=u3, am2=rowmap(3*r), K2=3*r, B2=Bd3, ldb2=3*r), 2.*M*d*(3*d+3*r), ["od"]),
    }
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
oa, ob = outs(), outs()
oa["o3"].copy_(rnd(M, 3 * d)); ob["o3"].copy_(oa["o3"])
ca, cb = cases(oa), cases(ob)
only = os.environ.get("ONLY")
for name in ca:
    if only and only not in name: continue
    fa, flops, keys = ca[name]; fb = cb[name][0]
    L.ns_debug_set_ring(4); fa(); torch.cuda.synchronize()
    L.ns_debug_set_ring(9); fb(); torch.cuda.synchronize()
    bad = {k: int((oa[k] != ob[k]).sum().item()) for k in keys}
    nan = {k: bool(torch.isnan(ob[k].float()).any().item()) for k in keys}
    best = {4: 1e9, 9: 1e9}
    if not os.environ.get("NOTIME"):
        for rep in range(4):
            for m, fn in ((4, fa), (9, fb)):
                L.ns_debug_set_ring(m)
                best[m] = min(best[m], t(fn))
    if name == "dgrad qkv+lora":   # o3 is an input there: keep both copies equal
        pass
    print(f"{name:22s} mismatches {bad} nan {nan}   one-tile {best[4]*1000:7.1f} us {flops/best[4]/1e9:6.0f} TF/s   persistent {best[9]*1000:7.1f} us {flops/best[9]/1e9:6.0f} TF/s", flush=True)
L.ns_debug_set_ring(1)
