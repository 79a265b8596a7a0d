"""Persistent (mode 1 default / 9) vs one-tile (mode 4) phase-interleaved GEMM on the conv-stem shapes of the bs64 step (segmented row maps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap, NS_GEMM_GELU, NS_GEMM_GELU_SAVE_GRAD, NS_GEMM_MUL_P16
dev = torch.device("cuda:0"); L = lib.load()
B, T, Cp, d = 64, 6000, 256, 512
T2, S = T // 2, T // 4
F16 = torch.float16
rnd = lambda *s, dtype=F16, scale=1.0: (torch.randn(*s, device=dev) * scale).to(dtype)
xin = rnd(B, T + 2, Cp); w0 = rnd(d, 3 * Cp, scale=.03); w1 = rnd(d, 3 * d, scale=.02); we = rnd(d, d, scale=.03); wo = rnd(d, 2 * d, scale=.03)
bias = rnd(d, dtype=torch.float32)
def mk():
    return dict(pre0=torch.zeros(B * T, d, device=dev, dtype=F16), g0=torch.zeros(B, T + 2, d, device=dev, dtype=F16),
                pre1=torch.zeros(B * T2, d, device=dev, dtype=F16), g1=torch.zeros(B, T2 + 2, d, device=dev, dtype=F16),
                dpre1=torch.zeros(B, T2 + 2, d, device=dev, dtype=F16), dpre0=torch.zeros(B * T, d, device=dev, dtype=F16))
dp2 = rnd(B, S + 2, d); dp1 = rnd(B, T2 + 2, d)
def cases(o):
    ev = rowmap(2 * d, S, T2 * d); evh = rowmap(2 * d, S, (T2 + 2) * d); ev0 = rowmap(2 * d, T2, T * d)
    return {
     "conv1.0+gelu": (lambda: ops.gemm(A=xin, am=rowmap(Cp, T, (T + 2) * Cp), K=3 * Cp, B=w0, ldb=3 * Cp, M=B * T, N=d, bias=bias, C16=o["pre0"], c16m=rowmap(d),
                                       G16=(o["g0"], d), g16m=rowmap(d, T, (T + 2) * d), flags=NS_GEMM_GELU | NS_GEMM_GELU_SAVE_GRAD), 2. * B * T * d * 3 * Cp, ["pre0", "g0"]),
     "conv1.2+gelu": (lambda: ops.gemm(A=o["g0"], am=rowmap(2 * d, T2, (T + 2) * d), K=3 * d, B=w1, ldb=3 * d, M=B * T2, N=d, bias=bias, C16=o["pre1"], c16m=rowmap(d),
                                       G16=(o["g1"], d), g16m=rowmap(d, T2, (T2 + 2) * d), flags=NS_GEMM_GELU | NS_GEMM_GELU_SAVE_GRAD), 2. * B * T2 * d * 3 * d, ["pre1", "g1"]),
     "dconv2 even": (lambda: ops.gemm(A=(dp2, d), am=rowmap(d, S, (S + 2) * d), K=d, B=we, ldb=d, M=B * S, N=d, C16=(o["dpre1"], d), c16m=evh, P16=o["pre1"], p16m=ev,
                                      flags=NS_GEMM_MUL_P16), 2. * B * S * d * d, ["dpre1"]),
     "dconv1.2 odd": (lambda: ops.gemm(A=(dp1, d), am=rowmap(d, T2, (T2 + 2) * d), K=2 * d, B=wo, ldb=2 * d, M=B * T2, N=d, C16=(o["dpre0"], d), c16m=ev0,
                                       P16=(o["pre0"], d), p16m=ev0, flags=NS_GEMM_MUL_P16), 2. * B * T2 * d * 2 * d, ["dpre0"]),
    }
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
oa, ob = mk(), mk()
ca, cb = cases(oa), cases(ob)
for name in ca:
    fa, flops, keys = ca[name]; fb = cb[name][0]
    L.ns_debug_set_ring(4); fa(); torch.cuda.synchronize()
    L.ns_debug_set_ring(9); fb(); torch.cuda.synchronize()
    bad = {k: int((oa[k] != ob[k]).sum().item()) for k in keys}
    best = {4: 1e9, 9: 1e9}
    for rep in range(4):
        for m, fn in ((4, fa), (9, fb)):
            L.ns_debug_set_ring(m)
            best[m] = min(best[m], t(fn))
    print(f"{name:16s} mismatches {bad}  one-tile {best[4]*1000:7.1f} us {flops/best[4]/1e9:6.0f} TF/s   persistent {best[9]*1000:7.1f} us {flops/best[9]/1e9:6.0f} TF/s", flush=True)
L.ns_debug_set_ring(1)
