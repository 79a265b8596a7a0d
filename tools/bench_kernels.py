"""Per-kernel micro-benchmarks at the whisper-base / B=64 shapes of the training step (GPU box only).
Usage: python tools/bench_kernels.py [filter-substring]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuspeech1_amd import ops  # noqa: E402
from neuspeech1_amd.ops import rowmap  # noqa: E402

dev = torch.device("cuda:0")
F16, F32 = torch.float16, torch.float32
flt = sys.argv[1] if len(sys.argv) > 1 else ""
if os.environ.get("NS_RING", "1") != "1":
    from neuspeech1_amd import lib as _l
    _l.load().ns_debug_set_ring(int(os.environ["NS_RING"]))
    print(f"[ns_debug_set_ring({os.environ['NS_RING']})]")


def rnd(*s, dtype=F16, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dtype)


def timeit(name, fn, flops=0.0, bytes_=0.0, iters=10):
    if flt and flt not in name:
        return
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:34s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TF/s  {bytes_ / ms / 1e6:8.1f} GB/s", flush=True)


B, S, d, f, H, r = 64, 1500, 512, 2048, 8, 32
M = B * S

x = rnd(M, d)
Wqkv, bqkv = rnd(3 * d, d, scale=0.04), rnd(3 * d, dtype=F32)
u3, sB3 = rnd(M, 3 * r), rnd(3 * d, r, scale=0.1)
qkv = torch.empty(M, 3 * d, device=dev, dtype=F16)
timeit("gemm qkv 96000x1536x512", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wqkv, ldb=d, M=M, N=3 * d, bias=bqkv, C16=qkv,
                                                    c16m=rowmap(3 * d)), 2.0 * M * 3 * d * d)
timeit("gemm qkv+lora", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wqkv, ldb=d, M=M, N=3 * d, bias=bqkv, C16=qkv,
                                         c16m=rowmap(3 * d), A2=u3, am2=rowmap(3 * r), K2=r, B2=sB3, ldb2=r, a2_ngroup=d),
       2.0 * M * 3 * d * (d + r))
duq, AqT = rnd(M, 3 * r), rnd(d, 3 * r, scale=0.1)
dx = torch.empty(M, d, device=dev, dtype=F16)
WqkvT = rnd(d, 3 * d, scale=0.04)
timeit("dgrad qkv 96000x512x1536", lambda: ops.gemm(A=qkv, am=rowmap(3 * d), K=3 * d, B=WqkvT, ldb=3 * d, M=M, N=d, C16=dx, c16m=rowmap(d)),
       2.0 * M * 3 * d * d)
timeit("dgrad qkv+lora(96)+drop", lambda: ops.gemm(A=qkv, am=rowmap(3 * d), K=3 * d, B=WqkvT, ldb=3 * d, M=M, N=d, C16=dx, c16m=rowmap(d),
                                                   A2=duq, am2=rowmap(3 * r), K2=3 * r, B2=AqT, ldb2=3 * r, drop_p=0.05, drop_seed=5),
       2.0 * M * (3 * d + 3 * r) * d)
Wo, bo = rnd(d, d, scale=0.04), rnd(d, dtype=F32)
h32 = rnd(M, d, dtype=F32)
h32o = torch.empty_like(h32)
timeit("gemm out+res 96000x512x512", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=Wo, ldb=d, M=M, N=d, bias=bo, R32=h32, H32=h32o,
                                                       h32m=rowmap(d)), 2.0 * M * d * d, M * d * (2 + 8))
W1, b1 = rnd(f, d, scale=0.04), rnd(f, dtype=F32)
pre, gf = torch.empty(M, f, device=dev, dtype=F16), torch.empty(M, f, device=dev, dtype=F16)
timeit("gemm fc1+gelu 96000x2048x512", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, bias=b1, C16=pre,
                                                         c16m=rowmap(f), G16=gf, g16m=rowmap(f), flags=ops.NS_GEMM_GELU),
       2.0 * M * d * f)
timeit("gemm fc1 plain (C16 only)", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, bias=b1, C16=pre, c16m=rowmap(f)),
       2.0 * M * d * f)
timeit("gemm fc1 gelu (G16 only)", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, bias=b1, G16=gf, g16m=rowmap(f),
                                                    flags=ops.NS_GEMM_GELU), 2.0 * M * d * f)
timeit("gemm fc1 gelu+savegrad+lora", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, bias=b1, C16=pre, c16m=rowmap(f),
                                                       G16=gf, g16m=rowmap(f), flags=ops.NS_GEMM_GELU | ops.NS_GEMM_GELU_SAVE_GRAD,
                                                       A2=u3, am2=rowmap(3 * r), K2=r, B2=rnd(f, r, scale=0.1), ldb2=r),
       2.0 * M * (d + r) * f)
W1T = rnd(d, f, scale=0.02)
du1, A1T = rnd(M, r), rnd(f, r, scale=0.1)
timeit("dgrad fc2 mulp 96000x2048x512", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, C16=gf, c16m=rowmap(f),
                                                         P16=pre, p16m=rowmap(f), flags=ops.NS_GEMM_MUL_P16), 2.0 * M * d * f)
timeit("dgrad fc2 mulp+lora+drop", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, C16=gf, c16m=rowmap(f),
                                                    P16=pre, p16m=rowmap(f), flags=ops.NS_GEMM_MUL_P16, A2=du1, am2=rowmap(r), K2=r,
                                                    B2=A1T, ldb2=r, drop_p=0.05, drop_seed=5), 2.0 * M * (d + r) * f)
W2 = rnd(d, f, scale=0.02)
timeit("gemm fc2+res 96000x512x2048", lambda: ops.gemm(A=gf, am=rowmap(f), K=f, B=W2, ldb=f, M=M, N=d, bias=bo, R32=h32,
                                                        H32=h32o, h32m=rowmap(d)), 2.0 * M * d * f)
timeit("gemm dgelu 96000x2048x512", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=W1, ldb=d, M=M, N=f, C16=gf, c16m=rowmap(f),
                                                      P16=pre, p16m=rowmap(f), flags=ops.NS_GEMM_DGELU), 2.0 * M * d * f)
A_l = rnd(3 * r, d, scale=0.04)
timeit("gemm skinny 96000x96x512", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=A_l, ldb=d, M=M, N=3 * r, C16=u3, c16m=rowmap(3 * r)),
       2.0 * M * 3 * r * d, M * d * 2)
timeit("gemm skinny+drop 96000x96x512", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=A_l, ldb=d, M=M, N=3 * r, C16=u3, c16m=rowmap(3 * r),
                                                          flags=ops.NS_GEMM_DROP_A, drop_p=0.05, drop_seed=7),
       2.0 * M * 3 * r * d, M * d * 2)
A2_l = rnd(r, f, scale=0.02)
u1 = torch.empty(M, r, device=dev, dtype=F16)
timeit("gemm skinny 96000x32x2048", lambda: ops.gemm(A=gf, am=rowmap(f), K=f, B=A2_l, ldb=f, M=M, N=r, C16=u1, c16m=rowmap(r)),
       2.0 * M * r * f, M * f * 2)
A1_l = rnd(r, d, scale=0.04)
timeit("gemm skinny+drop 96000x32x512", lambda: ops.gemm(A=x, am=rowmap(d), K=d, B=A1_l, ldb=d, M=M, N=r, C16=u1, c16m=rowmap(r),
                                                          flags=ops.NS_GEMM_DROP_A, drop_p=0.05, drop_seed=7),
       2.0 * M * r * d, M * d * 2)
timeit("gemm skinny+drop 96000x32x2048", lambda: ops.gemm(A=gf, am=rowmap(f), K=f, B=A2_l, ldb=f, M=M, N=r, C16=u1, c16m=rowmap(r),
                                                           flags=ops.NS_GEMM_DROP_A, drop_p=0.05, drop_seed=7),
       2.0 * M * r * f, M * f * 2)
# conv stem
T, Cp = 6000, 256
xin = rnd(B, T + 2, Cp)
wc0 = rnd(d, 3 * Cp, scale=0.04)
pre0 = torch.empty(B * T, d, device=dev, dtype=F16)
g0 = torch.zeros(B, T + 2, d, device=dev, dtype=F16)
timeit("conv1.0 384000x512x768", lambda: ops.gemm(A=xin, am=rowmap(Cp, T, (T + 2) * Cp), K=3 * Cp, B=wc0, ldb=3 * Cp, M=B * T, N=d,
                                                   bias=bo, C16=pre0, c16m=rowmap(d), G16=(g0, d), g16m=rowmap(d, T, (T + 2) * d),
                                                   flags=ops.NS_GEMM_GELU), 2.0 * B * T * d * 3 * Cp)
wc1 = rnd(d, 3 * d, scale=0.03)
pre1 = torch.empty(B * T // 2, d, device=dev, dtype=F16)
timeit("conv1.2 192000x512x1536", lambda: ops.gemm(A=g0, am=rowmap(2 * d, T // 2, (T + 2) * d), K=3 * d, B=wc1, ldb=3 * d, M=B * T // 2,
                                                    N=d, bias=bo, C16=pre1, c16m=rowmap(d)), 2.0 * B * T // 2 * d * 3 * d)
# LM head
ML, Vp = 64 * 44, 51968
xd, E = rnd(ML, d), rnd(Vp, d, scale=0.04)
logits = torch.empty(ML, Vp, device=dev, dtype=F16)
timeit("lmhead 2816x51968x512", lambda: ops.gemm(A=xd, am=rowmap(d), K=d, B=E, ldb=d, M=ML, N=Vp, C16=logits, c16m=rowmap(Vp)),
       2.0 * ML * Vp * d)
# TN weight gradients
gA = torch.zeros(r, f, device=dev, dtype=F32)
timeit("tn dA 32x2048 <- 96000", lambda: ops.gemm(A=u1, am=rowmap(r), K=M, B=gf, bm=rowmap(f), M=r, N=f, C32=gA, ldc32=f,
                                                   flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32, splits=48), 2.0 * M * r * f, M * f * 2)
gB = torch.zeros(d, r, device=dev, dtype=F32)
timeit("tn dB 512x32 <- 96000", lambda: ops.gemm(A=x, am=rowmap(d), K=M, B=u1, bm=rowmap(r), M=d, N=r, C32=gB, ldc32=r,
                                                  flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32, splits=192), 2.0 * M * r * d, M * d * 2)
gW = torch.zeros(d, 3 * Cp, device=dev, dtype=F32)
timeit("tn conv1.0 wgrad 512x768 <- 384000", lambda: ops.gemm(A=pre0, am=rowmap(d, T, T * d), K=B * T, B=xin,
                                                               bm=rowmap(Cp, T, (T + 2) * Cp), M=d, N=3 * Cp, C32=gW, ldc32=3 * Cp,
                                                               flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32, splits=32),
       2.0 * B * T * d * 3 * Cp)
# the bench's conv-stem weight gradients ("replace" front end): conv2 512 x 1536 <- 96 000 (stride 2, halo maps), conv1 512 x 768 <- 192 000
T2w = T // 2
dp2 = torch.zeros(B, S + 2, d, device=dev, dtype=F16); dp2.normal_()
g1h = torch.zeros(B, T2w + 2, d, device=dev, dtype=F16); g1h.normal_()
gW2, gb2 = torch.zeros(d, 3 * d, device=dev, dtype=F32), torch.zeros(d, device=dev, dtype=F32)
timeit("tn conv2 wgrad 512x1536 <- 96000", lambda: ops.gemm(A=(dp2, d), am=rowmap(d, S, (S + 2) * d), K=B * S, B=g1h,
                                                             bm=rowmap(2 * d, S, (T2w + 2) * d), M=d, N=3 * d, C32=gW2, ldc32=3 * d,
                                                             flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32 | ops.NS_GEMM_COLSUM_A, H32=gb2,
                                                             splits=16), 2.0 * B * S * d * 3 * d)
dp1 = torch.zeros(B, T2w + 2, d, device=dev, dtype=F16); dp1.normal_()
timeit("tn conv1 wgrad 512x768 <- 192000", lambda: ops.gemm(A=(dp1, d), am=rowmap(d, T2w, (T2w + 2) * d), K=B * T2w, B=xin,
                                                             bm=rowmap(2 * Cp, T2w, (T + 2) * Cp), M=d, N=3 * Cp, C32=gW, ldc32=3 * Cp,
                                                             flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32 | ops.NS_GEMM_COLSUM_A, H32=gb2,
                                                             splits=32), 2.0 * B * T2w * d * 3 * Cp)
# attention
ao = torch.empty(M, d, device=dev, dtype=F16)
lse = torch.empty(B, H, S, device=dev)
qkvr = rnd(M, 3 * d, scale=0.5)
common = dict(Q=qkvr, K=(qkvr, d), V=(qkvr, 2 * d), O=ao, B=B, H=H, Lq=S, Lk=S, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, causal=False,
              LSE=lse)
fl = 4.0 * B * H * S * S * 64
timeit("attn fwd S=1500", lambda: ops.attn_fwd(**common), fl)
dO, dqkv, delta = rnd(M, d), torch.empty(M, 3 * d, device=dev, dtype=F16), torch.empty(B, H, S, device=dev)
timeit("attn bwd S=1500", lambda: ops.attn_bwd(**common, dO=dO, dQ=dqkv, dK=(dqkv, d), dV=(dqkv, 2 * d), Delta=delta, lddo=d,
                                               lddq=3 * d, lddk=3 * d, lddv=3 * d), 2.5 * fl)
# byte movers
y16 = torch.empty(M, d, device=dev, dtype=F16)
mean, rstd, gam = torch.empty(M, device=dev), torch.empty(M, device=dev), torch.ones(d, device=dev)
timeit("layernorm fwd 96000x512", lambda: ops.layernorm_fwd(h32, gam, gam, y16, mean, rstd, M, d), 0, M * d * 6)
timeit("layernorm bwd 96000x512", lambda: ops.layernorm_bwd(y16, False, h32, mean, rstd, gam, h32, h32o, y16, M, d), 0, M * d * 16)
xs = rnd(B, 208, T, dtype=F32)
timeit("signal_pack 64x208x6000", lambda: ops.signal_pack(xs, xin, B, 208, T, Cp), 0, B * 208 * T * 4 + B * T * Cp * 2)

# conv bias gradients: column sums of d(pre) in halo layout
for name, rows_ in (("colsum 64x6000x512", 64 * 6000), ("colsum 64x3002x512", 64 * 3002)):
    a_ = rnd(rows_, 512)
    o_ = torch.zeros(512, device=dev)
    timeit(name, lambda: ops.colsum(a_, o_, rows_, 512, 512), 0.0, rows_ * 512 * 2)

# decoder-side GEMMs of the training step (ML = 64 x 44 label positions)
MLd = 64 * 44
xdd = rnd(MLd, d)
for name, n_, k_ in (("dec 2816x512x512", d, d), ("dec 2816x1536x512", 3 * d, d), ("dec 2816x2048x512", f, d), ("dec 2816x512x2048", d, f)):
    a_, w_ = rnd(MLd, k_), rnd(n_, k_, scale=0.04)
    o_ = torch.empty(MLd, n_, device=dev, dtype=F16)
    timeit(name, lambda: ops.gemm(A=a_, am=rowmap(k_), K=k_, B=w_, ldb=k_, M=MLd, N=n_, C16=o_, c16m=rowmap(n_)), 2.0 * MLd * n_ * k_)

# fused adapter backward (du + dB from one pass over dy): the three site shapes, workgroup count sweep
for name, G_, N_ in (("lora_bwd qkv G=3 N=512", 3, d), ("lora_bwd d-wide N=512", 1, d), ("lora_bwd fc1 N=2048", 1, f)):
    dy_ = rnd(M, G_ * N_, scale=0.5)
    u_ = rnd(M, G_ * r, scale=0.5)
    du_ = torch.empty(M, G_ * r, device=dev, dtype=F16)
    sBT_ = [rnd(r, N_, scale=0.1) for _ in range(G_)]
    dB_ = [torch.zeros(N_, r, device=dev) for _ in range(G_)]
    for sp, slabs in ((128, False), (256, False), (512, False), (256, True), (512, True)):
        timeit(f"{name} splits={sp} {'slabs' if slabs else 'atomics'}",
               lambda: ops.lora_bwd_dudb(dy=dy_, ldy=G_ * N_, u=u_, ldu=G_ * r, du=du_, lddu=G_ * r, sBT=sBT_, dB=dB_, lddb=r, M=M, N=N_, r=r,
                                         alpha_du=1.0, alpha_db=[1.0] * G_, splits=sp, slabs=slabs),
               0.0, M * G_ * N_ * 2)
