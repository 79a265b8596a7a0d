"""Per-kernel parity of libneuspeech_hip (through the C ABI) against plain torch
fp32 math on the same device.  Tolerances are stated per test: fp16 outputs
carry 2^-11 relative rounding, fp32 accumulation order differs from rocBLAS."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from neuspeech1_amd import ops as o
    return o


def rnd(shape, dev, scale=1.0, dtype=torch.float16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


def close(a, b, atol, rtol, what=""):
    a = a.float()
    b = b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{what}: {bad}/{err.numel()} off, max err {err.max().item():.4g} (ref max {b.abs().max().item():.4g})"


# --------------------------------------------------------------------------- GEMM
def test_gemm_integer_exact_layout(ops, dev):
    """A=I-style check with ASYMMETRIC integer data: catches transposed / permuted MFMA layouts exactly."""
    M, N, K = 256, 384, 128
    A = ((torch.arange(M * K, device=dev).reshape(M, K) * 7 + 3) % 5 - 2).half()
    B = ((torch.arange(N * K, device=dev).reshape(N, K) * 11 + 1) % 7 - 3).half()
    Cout = torch.zeros(M, N, device=dev, dtype=torch.float16)
    ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=Cout, c16m=ops.rowmap(N))
    ref = A.float() @ B.float().T
    assert torch.equal(Cout.float(), ref), f"max diff {(Cout.float() - ref).abs().max()}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 192), (1000, 1536, 512), (77, 96, 512), (513, 32, 2048)])
def test_gemm_nt_bias(ops, dev, M, N, K):
    A, B = rnd((M, K), dev, 1.0, seed=1), rnd((N, K), dev, 0.05, seed=2)
    bias = rnd((N,), dev, 0.5, torch.float32, seed=3)
    Cout = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
    ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, C16=Cout, c16m=ops.rowmap(N))
    ref = A.float() @ B.float().T + bias
    close(Cout, ref, 2e-3 * math.sqrt(K / 64), 2e-3, "gemm_nt")


@pytest.mark.parametrize("M,N,K", [(128, 512, 512), (128, 2048, 512), (128, 512, 2048), (640, 512, 528), (5, 520, 272),
                                   (33, 1284, 1280)])
def test_gemm_small_m_split_k_kernel(ops, dev, M, N, K):
    """decode shapes take the 32x32 split-K kernel (ns_gemm_smallm.hip): same results as the staged 128x32 tile
    (ns_debug_set_ring(6)) up to fp32 summation order, ragged M / N / K tails, GELU + in-place residual epilogues;
    integer data must come out exact through both"""
    from neuspeech1_amd import lib
    A, B = rnd((M, K), dev, 1.0, seed=1), rnd((N, K), dev, 0.05, seed=2)
    bias = rnd((N,), dev, 0.5, torch.float32, seed=3)
    R = rnd((M, N), dev, 1.0, torch.float32, seed=4)
    res = {}
    try:
        for mode in (6, 1):
            lib.load().ns_debug_set_ring(mode)
            C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
            Gl = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
            H = R.clone()
            ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, C16=C, c16m=ops.rowmap(N), G16=Gl,
                     g16m=ops.rowmap(N), flags=ops.NS_GEMM_GELU)
            ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, R32=H, H32=H, h32m=ops.rowmap(N))
            Ai = ((torch.arange(M * K, device=dev).reshape(M, K) * 7 + 3) % 5 - 2).half()
            Bi = ((torch.arange(N * K, device=dev).reshape(N, K) * 11 + 1) % 7 - 3).half()
            Ci = torch.zeros(M, N, device=dev, dtype=torch.float32)
            ops.gemm(A=Ai, am=ops.rowmap(K), K=K, B=Bi, ldb=K, M=M, N=N, C32=Ci, ldc32=N)
            assert torch.equal(Ci, Ai.float() @ Bi.float().T)
            res[mode] = (C, Gl, H)
    finally:
        lib.load().ns_debug_set_ring(1)
    ref = A.float() @ B.float().T + bias
    tol = 2e-3 * math.sqrt(K / 64)
    close(res[1][0], ref, tol, 2e-3, "C16")
    close(res[1][1], F.gelu(res[1][0].float()), 1e-3, 1e-3, "G16")
    close(res[1][2], R + ref.half().float(), 2 * tol, 2e-3, "H32 in place")
    close(res[1][0], res[6][0].float(), 2e-3, 2e-3, "small-M vs staged tile")


def test_gemm_epilogues(ops, dev):
    M, N, K, S = 384, 256, 128, 96
    A, B = rnd((M, K), dev, seed=1), rnd((N, K), dev, 0.1, seed=2)
    bias = rnd((N,), dev, 0.2, torch.float32, seed=3)
    R = rnd((M, N), dev, 1.0, torch.float32, seed=4)
    pos = rnd((S, N), dev, 1.0, torch.float32, seed=5)
    C16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    G16 = torch.empty_like(C16)
    H = torch.empty(M, N, device=dev, dtype=torch.float32)
    ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, C16=C16, c16m=ops.rowmap(N), G16=G16,
             g16m=ops.rowmap(N), R32=R, H32=H, h32m=ops.rowmap(N), pos=pos, pos_rows=S, flags=ops.NS_GEMM_GELU)
    v16 = (A.float() @ B.float().T + bias).half()
    close(C16, v16, 3e-3, 2e-3, "C16")
    g = F.gelu(C16.float()).half()  # gelu of the kernel's own rounded value: isolates the epilogue
    close(G16, g, 1e-3, 1e-3, "G16")
    href = R + G16.float() + pos.repeat(M // S, 1)
    close(H, href, 1e-5, 1e-6, "H32")
    # in-place residual, no gelu
    H2 = R.clone()
    ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, R32=H2, H32=H2, h32m=ops.rowmap(N))
    close(H2, R + v16.float(), 4e-3, 2e-3, "H32 inplace")
    # dgelu
    P = rnd((M, N), dev, 1.0, seed=6)
    D = torch.empty_like(C16)
    ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=D, c16m=ops.rowmap(N), P16=P, p16m=ops.rowmap(N),
             flags=ops.NS_GEMM_DGELU)
    x = P.float().requires_grad_(True)
    F.gelu(x).backward((A.float() @ B.float().T).half().float())
    close(D, x.grad, 4e-3, 3e-3, "dgelu")
    # gelu with the derivative saved in place of the pre-activation, then the multiply-only backward seam
    Dg = torch.empty_like(C16)
    G2 = torch.empty_like(C16)
    ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, C16=Dg, c16m=ops.rowmap(N), G16=G2,
             g16m=ops.rowmap(N), flags=ops.NS_GEMM_GELU | ops.NS_GEMM_GELU_SAVE_GRAD)
    xg = C16.float().requires_grad_(True)
    F.gelu(xg).sum().backward()
    close(Dg, xg.grad, 1e-3, 1e-3, "saved gelu'")
    assert torch.equal(G2, G16)
    D2 = torch.empty_like(C16)
    ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=D2, c16m=ops.rowmap(N), P16=Dg, p16m=ops.rowmap(N),
             flags=ops.NS_GEMM_MUL_P16)
    assert torch.equal(D2, ((A.float() @ B.float().T).half().float() * Dg.float()).half()) or \
        (D2.float() - (A.float() @ B.float().T).half().float() * Dg.float()).abs().max() < 4e-3


def test_gemm_second_product_groups(ops, dev):
    """fused q|k|v projection with three LoRA-A outputs side by side (a2_ngroup)."""
    M, d, r = 260, 256, 32
    x = rnd((M, d), dev, seed=1)
    W = rnd((3 * d, d), dev, 0.05, seed=2)
    u = rnd((M, 3 * r), dev, 0.5, seed=3)
    Bs = rnd((3 * d, r), dev, 0.1, seed=4)
    out = torch.empty(M, 3 * d, device=dev, dtype=torch.float16)
    ops.gemm(A=x, am=ops.rowmap(d), K=d, B=W, ldb=d, M=M, N=3 * d, A2=u, am2=ops.rowmap(3 * r), K2=r, B2=Bs, ldb2=r,
             a2_ngroup=d, C16=out, c16m=ops.rowmap(3 * d))
    ref = x.float() @ W.float().T
    for g in range(3):
        ref[:, g * d:(g + 1) * d] += u[:, g * r:(g + 1) * r].float() @ Bs[g * d:(g + 1) * d].float().T
    close(out, ref, 4e-3, 3e-3, "gemm 2nd product")
    # K2 = 16 (AdaLoRA r=12 padded) without groups
    u2, B2 = rnd((M, 16), dev, 0.5, seed=5), rnd((d, 16), dev, 0.1, seed=6)
    out2 = torch.empty(M, d, device=dev, dtype=torch.float16)
    ops.gemm(A=x, am=ops.rowmap(d), K=d, B=W, ldb=d, M=M, N=d, A2=u2, am2=ops.rowmap(16), K2=16, B2=B2, ldb2=16,
             C16=out2, c16m=ops.rowmap(d))
    close(out2, x.float() @ W[:d].float().T + u2.float() @ B2.float().T, 4e-3, 3e-3, "gemm K2=16")


@pytest.mark.parametrize("stride,T,Cin", [(1, 128, 64), (2, 256, 128), (2, 1000, 64)])
def test_gemm_conv_rowmap(ops, dev, stride, T, Cin):
    """k=3 Conv1d as ONE GEMM over the halo-padded token-major activation (overlapping rows)."""
    Bn, Cout = 3, 128
    Tout = T // stride
    x = rnd((Bn, Cin, T), dev, 1.0, torch.float32, seed=1)
    w = rnd((Cout, Cin, 3), dev, 0.1, torch.float32, seed=2)
    b = rnd((Cout,), dev, 0.1, torch.float32, seed=3)
    xh = torch.zeros(Bn, T + 2, Cin, device=dev, dtype=torch.float16)
    xh[:, 1:T + 1] = x.transpose(1, 2).half()
    wg = w.permute(0, 2, 1).reshape(Cout, 3 * Cin).half().contiguous()   # [n][(tap, c)]
    out = torch.zeros(Bn, Tout + 2, Cout, device=dev, dtype=torch.float16)
    ops.gemm(A=xh, am=ops.rowmap(stride * Cin, Tout, (T + 2) * Cin), K=3 * Cin, B=wg, ldb=3 * Cin, M=Bn * Tout,
             N=Cout, bias=b, C16=(out, Cout), c16m=ops.rowmap(Cout, Tout, (Tout + 2) * Cout))
    ref = F.conv1d(xh[:, 1:T + 1].float().transpose(1, 2), wg.float().reshape(Cout, 3, Cin).permute(0, 2, 1), b,
                   stride=stride, padding=1).transpose(1, 2)
    close(out[:, 1:Tout + 1], ref, 5e-3, 3e-3, "conv-as-gemm")
    assert out[:, 0].abs().max() == 0 and out[:, Tout + 1].abs().max() == 0, "halo rows must stay zero"


@pytest.mark.parametrize("Mred,No,Ko,splits", [(512, 128, 128, 1), (1000, 256, 96, 4), (3000, 64, 32, 7), (4096, 96, 512, 16)])
def test_gemm_tn(ops, dev, Mred, No, Ko, splits):
    dY, X = rnd((Mred, No), dev, 0.5, seed=1), rnd((Mred, Ko), dev, 0.5, seed=2)
    out = torch.zeros(No, Ko, device=dev, dtype=torch.float32)
    ops.gemm(A=dY, am=ops.rowmap(No), K=Mred, B=X, bm=ops.rowmap(Ko), M=No, N=Ko, C32=out, ldc32=Ko,
             flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32, splits=splits, alpha=2.0)
    ref = 2.0 * dY.float().T @ X.float()
    close(out, ref, 2e-2, 2e-3, "gemm_tn")


def test_gemm_tn_conv_wgrad(ops, dev):
    """weight gradient of the stride-2 conv = dY^T * overlapping-row view, segmented per batch item."""
    Bn, T, Cin, Cout, stride = 3, 200, 64, 128, 2
    Tout = T // stride
    xh = torch.zeros(Bn, T + 2, Cin, device=dev, dtype=torch.float16)
    xh[:, 1:T + 1] = rnd((Bn, T, Cin), dev, 1.0, seed=1)
    dy = rnd((Bn * Tout, Cout), dev, 0.5, seed=2)
    out = torch.zeros(Cout, 3 * Cin, device=dev, dtype=torch.float32)
    ops.gemm(A=dy, am=ops.rowmap(Cout, Tout, Tout * Cout), K=Bn * Tout, B=xh,
             bm=ops.rowmap(stride * Cin, Tout, (T + 2) * Cin), M=Cout, N=3 * Cin, C32=out, ldc32=3 * Cin,
             flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32, splits=3)
    w = torch.zeros(Cout, Cin, 3, device=dev, requires_grad=True)
    y = F.conv1d(xh[:, 1:T + 1].float().transpose(1, 2), w, None, stride=stride, padding=1)
    y.backward(dy.float().reshape(Bn, Tout, Cout).transpose(1, 2))
    ref = w.grad.permute(0, 2, 1).reshape(Cout, 3 * Cin)
    close(out, ref, 2e-2, 2e-3, "conv wgrad")
    # the bias gradient (column sums of dy) as a side output of the same pass, accumulated onto a non-zero buffer
    out2 = torch.zeros_like(out)
    bg = torch.full((Cout,), 0.5, device=dev)
    ops.gemm(A=dy, am=ops.rowmap(Cout, Tout, Tout * Cout), K=Bn * Tout, B=xh,
             bm=ops.rowmap(stride * Cin, Tout, (T + 2) * Cin), M=Cout, N=3 * Cin, C32=out2, ldc32=3 * Cin,
             flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32 | ops.NS_GEMM_COLSUM_A, splits=3, H32=bg)
    close(out2, ref, 2e-2, 2e-3, "conv wgrad (with column sums)")
    close(bg, 0.5 + dy.float().sum(0), 2e-3, 1e-3, "bias gradient from the weight-gradient pass")


@pytest.mark.parametrize("Bn,T,Cin,Cout,stride", [(9, 4000, 256, 512, 2), (11, 1500, 512, 512, 1), (7, 6000, 256, 256, 2),
                                                  (5, 6004, 320, 512, 1)])   # 273 channels padded to 320: N = 960, a ragged last tile
def test_gemm_tn_conv_wgrad_256_tiles(ops, dev, Bn, T, Cin, Cout, stride):
    """ns_gemm_tn256 (256 x 256 LDS-DMA tiles, reduction splits chosen by the launcher) on conv-stem weight-gradient shapes:
    halo row maps of both operands, stride 1 and 2, a reduction length that is not a multiple of the stage depth or of the
    split count, accumulation onto a non-zero C32, the bias-gradient side output; against conv1d autograd in fp32 and
    against the 128 x 128 kernel it replaces."""
    from neuspeech1_amd import lib
    Tout = T // stride
    xh = torch.zeros(Bn, T + 2, Cin, device=dev, dtype=torch.float16)
    xh[:, 1:T + 1] = rnd((Bn, T, Cin), dev, 1.0, seed=1)
    dy = rnd((Bn * Tout, Cout), dev, 0.5, seed=2)
    assert Bn * Tout >= 16384 and Bn * Tout % 32 != 0
    w = torch.zeros(Cout, Cin, 3, device=dev, requires_grad=True)
    y = F.conv1d(xh[:, 1:T + 1].float().transpose(1, 2), w, None, stride=stride, padding=1)
    y.backward(dy.float().reshape(Bn, Tout, Cout).transpose(1, 2))
    ref = w.grad.permute(0, 2, 1).reshape(Cout, 3 * Cin)
    outs = []
    for mode in (1, 8):
        lib.load().ns_debug_set_ring(mode)
        out = torch.full((Cout, 3 * Cin), 0.25, device=dev, dtype=torch.float32)
        bg = torch.full((Cout,), 0.5, device=dev)
        ops.gemm(A=dy, am=ops.rowmap(Cout, Tout, Tout * Cout), K=Bn * Tout, B=xh,
                 bm=ops.rowmap(stride * Cin, Tout, (T + 2) * Cin), M=Cout, N=3 * Cin, C32=out, ldc32=3 * Cin,
                 flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32 | ops.NS_GEMM_COLSUM_A, splits=16, H32=bg, alpha=0.5)
        outs.append((out, bg))
    lib.load().ns_debug_set_ring(1)
    scale = (Bn * Tout / 300) ** 0.5
    close(outs[0][0], 0.25 + 0.5 * ref, 2e-2 * scale, 2e-3, "conv wgrad (256 x 256 tiles)")
    close(outs[0][1], 0.5 + 0.5 * dy.float().sum(0), 2e-3 * scale, 1e-3, "bias gradient (256 x 256 tiles)")
    close(outs[0][0], outs[1][0], 2e-3 * scale, 1e-4, "256 x 256 tiles vs 128 x 128 tiles")
    # plain row maps (no segments)
    a2, b2 = rnd((20000, 256), dev, 0.5, seed=5), rnd((20000, 512), dev, 0.5, seed=6)
    c2 = torch.zeros(256, 512, device=dev)
    ops.gemm(A=a2, am=ops.rowmap(256), K=20000, B=b2, bm=ops.rowmap(512), M=256, N=512, C32=c2, ldc32=512,
             flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32, splits=4)
    close(c2, a2.float().T @ b2.float(), 2e-1, 2e-3, "plain TN through the 256 x 256 tiles")
    # large-v2 widths: 5 x 15 tiles, three reduction splits
    if Cin == 512:
        a3, b3 = rnd((16500, 1280), dev, 0.5, seed=7), rnd((16500, 3840), dev, 0.5, seed=8)
        c3 = torch.zeros(1280, 3840, device=dev)
        ops.gemm(A=a3, am=ops.rowmap(1280), K=16500, B=b3, bm=ops.rowmap(3840), M=1280, N=3840, C32=c3, ldc32=3840,
                 flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32, splits=4)
        close(c3, a3.float().T @ b3.float(), 2e-1, 2e-3, "75 tiles of 256 x 256")


@pytest.mark.parametrize("M,p", [(12000, 0.05), (12031, 0.0)])
def test_gelu_epilogue_side_product(ops, dev, M, p):
    """ns_gemm_desc.side_*: the large-M GELU GEMM (fc1) also leaves, per 256-column tile, the product of its masked fp16 GELU
    output with a (32 x N) matrix (the next Linear's LoRA down-projection); ns_gemm_side_reduce sums the slabs.  Against torch
    on the kernel's own GELU output, and against the stand-alone down-projection kernel; the main outputs must be unchanged."""
    N, K, r, seed = 2048, 512, 32, 99
    assert ops.gemm_side_supported(M, N, K) and not ops.gemm_side_supported(1000, N, K) and not ops.gemm_side_supported(M, N + 64, K)
    x = rnd((M, K), dev, 1.0, seed=1)
    W = rnd((N, K), dev, 0.05, seed=2)
    bias = rnd((N,), dev, 0.2, torch.float32, seed=3)
    A2 = rnd((r, N), dev, 0.05, seed=4)
    pre, gf = torch.empty(M, N, device=dev, dtype=torch.float16), torch.empty(M, N, device=dev, dtype=torch.float16)
    slabs = torch.full(((N // 256) * M * 32,), float("nan"), device=dev)
    flags = ops.NS_GEMM_GELU | ops.NS_GEMM_GELU_SAVE_GRAD
    ops.gemm(A=x, am=ops.rowmap(K), K=K, B=W, ldb=K, M=M, N=N, bias=bias, C16=pre, c16m=ops.rowmap(N), G16=gf, g16m=ops.rowmap(N),
             flags=flags, side_B=A2, side_ldb=N, side_n=r, side_out=slabs, side_drop_p=p, side_drop_seed=seed)
    keep, inv = _keep_mask(seed, M, N, p, dev) if p > 0 else (torch.ones(M, N, device=dev), 1.0)
    u = torch.full((M, r + 8), float("nan"), device=dev, dtype=torch.float16)
    ops.gemm_side_reduce(slabs, N // 256, M, inv, u, r + 8)
    assert not torch.isnan(slabs).any() and torch.isnan(u[:, r:].float()).all()
    ref = ((gf.float() * keep) @ A2.float().T) * inv
    close(u[:, :r], ref, 2e-2, 5e-3, "side product vs torch on the kernel's GELU output")
    # the stand-alone kernel it replaces
    u2 = torch.empty(M, r, device=dev, dtype=torch.float16)
    ops.gemm(A=gf, am=ops.rowmap(N), K=N, B=A2, ldb=N, M=M, N=r, C16=u2, c16m=ops.rowmap(r), flags=ops.NS_GEMM_DROP_A if p > 0 else 0,
             alpha=inv, drop_p=p, drop_seed=seed)
    close(u[:, :r], u2, 1e-2, 5e-3, "side product vs the down-projection kernel")
    # main outputs unchanged by the side product
    pre0, gf0 = torch.empty_like(pre), torch.empty_like(gf)
    ops.gemm(A=x, am=ops.rowmap(K), K=K, B=W, ldb=K, M=M, N=N, bias=bias, C16=pre0, c16m=ops.rowmap(N), G16=gf0, g16m=ops.rowmap(N), flags=flags)
    assert torch.equal(pre, pre0) and torch.equal(gf, gf0)


# --------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,d", [(37, 256), (1000, 512), (130, 1280)])
def test_layernorm(ops, dev, rows, d):
    x = rnd((rows, d), dev, 2.0, torch.float32, seed=1) + 0.5
    g = rnd((d,), dev, 1.0, torch.float32, seed=2)
    b = rnd((d,), dev, 1.0, torch.float32, seed=3)
    y16 = torch.empty(rows, d, device=dev, dtype=torch.float16)
    y32 = torch.empty(rows, d, device=dev, dtype=torch.float32)
    mean = torch.empty(rows, device=dev)
    rstd = torch.empty(rows, device=dev)
    ops.layernorm_fwd(x, g, b, y16, mean, rstd, rows, d, y32=y32)
    xr = x.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (d,), g, b, 1e-5)
    close(y32, ref, 2e-5, 1e-5, "ln fwd32")
    close(y16, ref.half(), 1e-3, 1e-3, "ln fwd16")
    for is32 in (False, True):
        dy = rnd((rows, d), dev, 1.0, torch.float32 if is32 else torch.float16, seed=4)
        dres = rnd((rows, d), dev, 1.0, torch.float32, seed=5)
        dx32 = torch.empty(rows, d, device=dev)
        dx16 = torch.empty(rows, d, device=dev, dtype=torch.float16)
        ops.layernorm_bwd(dy, is32, x, mean, rstd, g, dres, dx32, dx16, rows, d)
        xr.grad = None
        ref.backward(dy.float(), retain_graph=True)
        close(dx32, xr.grad + dres, 3e-4, 1e-4, "ln bwd")
        close(dx16, dx32.half(), 1e-6, 1e-6, "ln bwd16")


# --------------------------------------------------------------------------- byte movers
@pytest.mark.parametrize("pad", [8, 64])
@pytest.mark.parametrize("ch,T", [(208, 6000), (273, 6000), (10, 100), (20, 402), (70, 130)])
def test_signal_pack(ops, dev, ch, T, pad):
    """Cp = the next multiple of 8 (what WhisperDims.ch_pad gives: the last 64-channel block of the kernel's grid is partial) and of 64"""
    Bn = 2
    Cp = (ch + pad - 1) // pad * pad
    x = rnd((Bn, ch, T), dev, 0.35, torch.float32, seed=1).clamp(-1, 1)
    out = torch.full((Bn, T + 2, Cp), float("nan"), device=dev, dtype=torch.float16)
    ops.signal_pack(x, out, Bn, ch, T, Cp)
    ref = torch.zeros(Bn, T + 2, Cp, device=dev, dtype=torch.float16)
    ref[:, 1:T + 1, :ch] = x.transpose(1, 2).half()
    assert torch.equal(out, ref)


def test_embed_cast_dgelu_colsum(ops, dev):
    V, d, Bn, Lq = 500, 256, 3, 7
    E, P = rnd((V, d), dev, 1.0, torch.float32, seed=1), rnd((448, d), dev, 1.0, torch.float32, seed=2)
    ids = torch.randint(0, V, (Bn, Lq), device=dev)
    h = torch.empty(Bn * Lq, d, device=dev)
    ops.embed_pos(ids, E, P, h, Bn * Lq, Lq, d, pos0=3)
    assert torch.equal(h.reshape(Bn, Lq, d), E[ids] + P[3:3 + Lq])
    # cast jobs
    src = rnd((70, 45), dev, 1.0, torch.float32, seed=3)
    d1 = torch.zeros(70, 48, device=dev, dtype=torch.float16)
    d2 = torch.zeros(45, 72, device=dev, dtype=torch.float16)
    tab, n = ops.make_cast_jobs([(src.data_ptr(), d1.data_ptr(), 70, 45, 45, 48, 2.0, 0),
                                 (src.data_ptr(), d2.data_ptr(), 70, 45, 45, 72, 0.5, 1)], dev)
    ops.cast_jobs(tab, n)
    assert torch.equal(d1[:, :45], (2.0 * src).half()) and torch.equal(d2[:, :70], (0.5 * src).half().T)
    # dgelu_mul into a halo layout + colsum
    rows, cols, seg = 128, 64, 64
    a, pre = rnd((rows, cols), dev, 1.0, seed=4), rnd((rows, cols), dev, 1.0, seed=5)
    out = torch.zeros(2, seg + 2, cols, device=dev, dtype=torch.float16)
    out_m = torch.zeros(2, seg + 2, cols, device=dev, dtype=torch.float16)
    ops.dgelu_mul(a, pre, (out_m, cols), ops.rowmap(cols, seg, (seg + 2) * cols), rows, cols, pre_is_grad=True)
    assert torch.equal(out_m[:, 1:seg + 1].reshape(rows, cols), (a.float() * pre.float()).half())
    ops.dgelu_mul(a, pre, (out, cols), ops.rowmap(cols, seg, (seg + 2) * cols), rows, cols)
    x = pre.float().requires_grad_(True)
    F.gelu(x).backward(a.float())
    close(out[:, 1:seg + 1].reshape(rows, cols), x.grad, 2e-3, 2e-3, "dgelu_mul")
    cs = torch.zeros(cols, device=dev)
    ops.colsum(a, cs, rows, cols, cols, 0.5)
    close(cs, 0.5 * a.float().sum(0), 1e-3, 1e-4, "colsum")


# --------------------------------------------------------------------------- attention
def attn_ref(q, k, v, causal):
    # q,k,v: (B, L, H, 64) fp32 ; returns (B, Lq, H, 64)
    s = torch.einsum("bqhd,bkhd->bhqk", q, k)
    if causal:
        Lq, Lk = q.shape[1], k.shape[1]
        i = torch.arange(Lq, device=q.device)[:, None]
        j = torch.arange(Lk, device=q.device)[None, :]
        s = s.masked_fill(j > i + (Lk - Lq), float("-inf"))
    p = s.softmax(-1)
    return torch.einsum("bhqk,bkhd->bqhd", p, v), torch.logsumexp(s, -1)


@pytest.mark.parametrize("Bn,H,Lq,Lk,causal", [(2, 4, 1500, 1500, False), (3, 2, 40, 40, True), (2, 4, 37, 1500, False),
                                                (2, 2, 1, 29, True), (1, 2, 200, 200, True)])
def test_attention_fwd_bwd(ops, dev, Bn, H, Lq, Lk, causal):
    d = H * 64
    # q lives in a fused (q|k|v)-style buffer to exercise row strides
    qkv = rnd((Bn * Lq, 3 * d), dev, 0.3, seed=1)
    kv = rnd((Bn * Lk, 2 * d), dev, 0.6, seed=2)
    O = torch.zeros(Bn * Lq, d, device=dev, dtype=torch.float16)
    LSE = torch.zeros(Bn, H, Lq, device=dev)
    common = dict(Q=qkv, K=kv, V=(kv, d), O=O, B=Bn, H=H, Lq=Lq, Lk=Lk, ldq=3 * d, ldk=2 * d, ldv=2 * d, ldo=d,
                  causal=causal, LSE=LSE)
    ops.attn_fwd(**common)
    q = qkv[:, :d].float().reshape(Bn, Lq, H, 64).requires_grad_(True)
    k = kv[:, :d].float().reshape(Bn, Lk, H, 64).requires_grad_(True)
    v = kv[:, d:].float().reshape(Bn, Lk, H, 64).requires_grad_(True)
    ref, lse_ref = attn_ref(q, k, v, causal)
    close(O.reshape(Bn, Lq, H, 64), ref, 3e-3, 1e-2, "attn fwd")
    close(LSE, lse_ref, 2e-3, 1e-3, "lse")
    dO = rnd((Bn * Lq, d), dev, 0.5, seed=3)
    dQ = torch.zeros(Bn * Lq, 3 * d, device=dev, dtype=torch.float16)
    dKV = torch.zeros(Bn * Lk, 2 * d, device=dev, dtype=torch.float16)
    Delta = torch.zeros(Bn, H, Lq, device=dev)
    ops.attn_bwd(**common, dO=dO, dQ=dQ, dK=dKV, dV=(dKV, d), Delta=Delta, lddo=d, lddq=3 * d, lddk=2 * d, lddv=2 * d)
    ref.backward(dO.float().reshape(Bn, Lq, H, 64))
    sc = max(1.0, math.sqrt(Lq / 64))
    close(dQ[:, :d].reshape(Bn, Lq, H, 64), q.grad, 4e-3, 2e-2, "dQ")
    close(dKV[:, :d].reshape(Bn, Lk, H, 64), k.grad, 4e-3 * sc, 2e-2, "dK")
    close(dKV[:, d:].reshape(Bn, Lk, H, 64), v.grad, 4e-3 * sc, 2e-2, "dV")
    assert dQ[:, d:].abs().max() == 0, "attention must not write outside its head columns"


@pytest.mark.parametrize("Bn,H,Lq,Lk,scale", [(3, 2, 44, 1500, 0.3), (1, 3, 64, 300, 0.3), (2, 1, 1, 257, 0.3), (2, 2, 45, 1281, 1.0), (64, 8, 44, 1500, 0.3)])
def test_attention_backward_few_queries_one_pass(ops, dev, Bn, H, Lq, Lk, scale):
    """ns_attn_bwd with a workspace at <= 64 queries (the decoder's cross-attention: label length x 1500 encoder states) = attn_bwd_fewq_kernel:
    K / V streamed once, dK / dV as in the two-pass kernel (bit-identical), dQ from the transposed recomputation through fp32 slabs (one per group
    of key blocks) + a fixed-order reduction.  Against torch autograd in fp32 and against the two-pass kernels; ragged key counts (300, 257 = one key in
    the last block of its group, 1281 = one key in a group of its own), one query, a full 64-query tile, and the bench size."""
    d = H * 64
    qx = rnd((Bn * Lq, d), dev, scale, seed=1)
    kv = rnd((Bn * Lk, 2 * d), dev, 2 * scale, seed=2)
    O = torch.zeros(Bn * Lq, d, device=dev, dtype=torch.float16)
    LSE = torch.zeros(Bn, H, Lq, device=dev)
    common = dict(Q=qx, K=kv, V=(kv, d), O=O, B=Bn, H=H, Lq=Lq, Lk=Lk, ldq=d, ldk=2 * d, ldv=2 * d, ldo=d, causal=False, LSE=LSE)
    ops.attn_fwd(**common)
    dO = rnd((Bn * Lq, d), dev, 0.5, seed=3)
    nws = ops.attn_bwd_workspace_bytes(Bn, H, Lq, Lk)
    assert nws % (Bn * H * Lq * 64 * 4) == 0 and nws > 0      # one fp32 dQ slab per group of key blocks
    res = {}
    for name, ws in (("two", None), ("one", torch.full((nws,), 0xFF, device=dev, dtype=torch.uint8))):   # slabs start as NaNs
        dQ = torch.full((Bn * Lq, d), float("nan"), device=dev, dtype=torch.float16)
        dKV = torch.full((Bn * Lk, 2 * d), float("nan"), device=dev, dtype=torch.float16)
        Delta = torch.zeros(Bn, H, Lq, device=dev)
        ops.attn_bwd(**common, dO=dO, dQ=dQ, dK=dKV, dV=(dKV, d), Delta=Delta, lddo=d, lddq=d, lddk=2 * d, lddv=2 * d, workspace=ws)
        assert not torch.isnan(dQ.float()).any() and not torch.isnan(dKV.float()).any(), name
        res[name] = (dQ.reshape(Bn, Lq, H, 64).float(), dKV[:, :d].reshape(Bn, Lk, H, 64).float(), dKV[:, d:].reshape(Bn, Lk, H, 64).float(), Delta.clone())
    dq1, dk1, dv1, de1 = res["one"]
    dq2, dk2, dv2, de2 = res["two"]
    close(de1, de2, 1e-5, 1e-5, "delta")
    # dK / dV: the same products in the same order as the two-pass kernel, from a delta summed in another order (last-bit differences in dS)
    close(dk1, dk2, 2e-3 * max(dk2.std().item(), 1e-3), 4e-3, "dK one pass vs two passes")
    close(dv1, dv2, 2e-3 * max(dv2.std().item(), 1e-3), 4e-3, "dV one pass vs two passes")
    if Bn * H <= 16:     # autograd reference (the bench-size case compares the two kernel forms only)
        q = qx.float().reshape(Bn, Lq, H, 64).requires_grad_(True)
        k = kv[:, :d].float().reshape(Bn, Lk, H, 64).requires_grad_(True)
        v = kv[:, d:].float().reshape(Bn, Lk, H, 64).requires_grad_(True)
        ref, _ = attn_ref(q, k, v, False)
        ref.backward(dO.float().reshape(Bn, Lq, H, 64))
        close(dq1, q.grad, max(4e-3, 2e-2 * q.grad.std().item()), 2e-2, "dQ")
        # whole-tensor relative error against autograd: no worse than the two-pass kernels (element-wise bounds on dK depend on how sharp the softmax is)
        rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
        for nm, one, two, g in (("dQ", dq1, dq2, q.grad), ("dK", dk1, dk2, k.grad), ("dV", dv1, dv2, v.grad)):
            e1, e2 = rel(one, g), rel(two, g)
            print(f"  [{Lq}x{Lk} scale {scale}] {nm}: one-pass rel {e1:.2e}, two-pass rel {e2:.2e}")
            assert e1 < 1.2 * e2 + 2e-4 and e1 < 2e-2, (nm, e1, e2)
    else:
        close(dq1, dq2, 2e-2 * dq2.std().item(), 2e-2, "dQ one pass vs two passes")
    # bitwise reproducible: fixed-order slab sums, no atomics
    dQb = torch.zeros(Bn * Lq, d, device=dev, dtype=torch.float16)
    dKVb = torch.zeros(Bn * Lk, 2 * d, device=dev, dtype=torch.float16)
    ops.attn_bwd(**common, dO=dO, dQ=dQb, dK=dKVb, dV=(dKVb, d), Delta=torch.zeros(Bn, H, Lq, device=dev), lddo=d, lddq=d, lddk=2 * d, lddv=2 * d,
                 workspace=torch.zeros(nws, device=dev, dtype=torch.uint8))
    assert torch.equal(dQb.reshape(Bn, Lq, H, 64).float(), dq1)


@pytest.mark.parametrize("Bn,H,Lq,Lk,scale", [(2, 4, 1500, 1500, 0.3), (1, 2, 700, 700, 0.3), (1, 3, 256, 512, 0.3), (2, 2, 1500, 1500, 1.2),
                                               (1, 1, 320, 1281, 0.3)])
def test_attention_backward_one_pass(ops, dev, Bn, H, Lq, Lk, scale):
    """ns_attn_bwd with a workspace = the ONE-pass backward (csrc/ns_attn_bwd1.hip: S and dP formed once, dQ summed over
    the key sweeps in fp32 scratch) against torch autograd in fp32 and against the two-pass kernels, on ragged key / query
    counts (700 = 2 x 256 + 188 keys, 10 x 64 + 60 queries; 1281 keys: a last sweep with ONE valid key) and with large
    logits (scale 1.2: |S| up to ~40, the regime where a rounded log2(e) on the operand would show)."""
    d = H * 64
    qkv = rnd((Bn * Lq, 3 * d), dev, scale, seed=1)
    kv = rnd((Bn * Lk, 2 * d), dev, 2 * scale, seed=2)
    O = torch.zeros(Bn * Lq, d, device=dev, dtype=torch.float16)
    LSE = torch.zeros(Bn, H, Lq, device=dev)
    common = dict(Q=qkv, K=kv, V=(kv, d), O=O, B=Bn, H=H, Lq=Lq, Lk=Lk, ldq=3 * d, ldk=2 * d, ldv=2 * d, ldo=d, causal=False, LSE=LSE)
    ops.attn_fwd(**common)
    q = qkv[:, :d].float().reshape(Bn, Lq, H, 64).requires_grad_(True)
    k = kv[:, :d].float().reshape(Bn, Lk, H, 64).requires_grad_(True)
    v = kv[:, d:].float().reshape(Bn, Lk, H, 64).requires_grad_(True)
    ref, _ = attn_ref(q, k, v, False)
    dO = rnd((Bn * Lq, d), dev, 0.5, seed=3)
    ref.backward(dO.float().reshape(Bn, Lq, H, 64))
    nws = ops.attn_bwd_workspace_bytes(Bn, H, Lq, Lk)
    assert nws == Bn * H * ((Lq + 63) // 64) * 64 * 64 * 4
    assert ops.attn_bwd_workspace_bytes(Bn, H, 100, Lk) == 0 and ops.attn_bwd_workspace_bytes(Bn, H, Lq, Lk, causal=True) == 0
    res = {}
    for name, ws in (("two", None), ("one", torch.full((nws,), 0xFF, device=dev, dtype=torch.uint8))):   # scratch starts as NaNs
        dQ = torch.zeros(Bn * Lq, 3 * d, device=dev, dtype=torch.float16)
        dKV = torch.zeros(Bn * Lk, 2 * d, device=dev, dtype=torch.float16)
        Delta = torch.zeros(Bn, H, Lq, device=dev)
        ops.attn_bwd(**common, dO=dO, dQ=dQ, dK=dKV, dV=(dKV, d), Delta=Delta, lddo=d, lddq=3 * d, lddk=2 * d, lddv=2 * d, workspace=ws)
        assert dQ[:, d:].abs().max() == 0, "attention must not write outside its head columns"
        res[name] = (dQ[:, :d].reshape(Bn, Lq, H, 64).float(), dKV[:, :d].reshape(Bn, Lk, H, 64).float(),
                     dKV[:, d:].reshape(Bn, Lk, H, 64).float(), Delta.clone())
    sc = max(1.0, math.sqrt(Lq / 64))
    dq1, dk1, dv1, de1 = res["one"]
    # element-wise: an absolute part of 2 % of the tensor's spread (fp16 rounding of dS scales with the values it sums)
    close(dq1, q.grad, max(4e-3, 2e-2 * q.grad.std().item()), 2e-2, "dQ")
    close(dk1, k.grad, max(4e-3 * sc, 2e-2 * k.grad.std().item()), 2e-2, "dK")
    close(dv1, v.grad, max(4e-3 * sc, 2e-2 * v.grad.std().item()), 2e-2, "dV")
    close(de1, res["two"][3], 1e-5, 1e-5, "delta")
    # and it is at least as close to autograd as the two-pass form (whole-tensor relative error)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
    for nm, one, two, g in (("dQ", dq1, res["two"][0], q.grad), ("dK", dk1, res["two"][1], k.grad), ("dV", dv1, res["two"][2], v.grad)):
        e1, e2 = rel(one, g), rel(two, g)
        print(f"  [{Lq}x{Lk} scale {scale}] {nm}: one-pass rel {e1:.2e}, two-pass rel {e2:.2e}")
        assert e1 < 1.2 * e2 + 2e-4, (nm, e1, e2)
    # bitwise reproducible: no atomics anywhere in the one-pass form
    dQ2 = torch.zeros(Bn * Lq, 3 * d, device=dev, dtype=torch.float16)
    dKV2 = torch.zeros(Bn * Lk, 2 * d, device=dev, dtype=torch.float16)
    ops.attn_bwd(**common, dO=dO, dQ=dQ2, dK=dKV2, dV=(dKV2, d), Delta=torch.zeros(Bn, H, Lq, device=dev), lddo=d, lddq=3 * d,
                 lddk=2 * d, lddv=2 * d, workspace=torch.zeros(nws, device=dev, dtype=torch.uint8))
    assert torch.equal(dQ2[:, :d].reshape(Bn, Lq, H, 64).float(), dq1) and torch.equal(dKV2[:, :d].reshape(Bn, Lk, H, 64).float(), dk1)


# --------------------------------------------------------------------------- loss / optimizer
def test_cross_entropy_and_argmax(ops, dev):
    rows, V, ldv = 50, 51865, 51968
    logits = torch.zeros(rows, ldv, device=dev, dtype=torch.float16)
    logits[:, :V] = rnd((rows, V), dev, 2.0, seed=1)
    logits[:, V:] = 7.0  # padding columns must be ignored
    labels = torch.randint(0, V, (rows,), device=dev)
    labels[::7] = -100
    row_loss = torch.empty(rows, device=dev)
    dl = torch.empty_like(logits)
    nvalid = torch.zeros(1, device=dev, dtype=torch.int32)
    loss = torch.zeros(1, device=dev)
    scale = torch.tensor([1024.0], device=dev)
    ops.cross_entropy(logits, labels, rows, V, ldv, row_loss, dl, nvalid, scale, loss)
    x = logits[:, :V].float().requires_grad_(True)
    ref = F.cross_entropy(x, labels, ignore_index=-100)
    (ref * 1024.0).backward()
    assert nvalid.item() == (labels != -100).sum().item()
    close(loss, ref.reshape(1), 1e-4, 1e-5, "loss")
    close(dl[:, :V], x.grad, 2e-3, 2e-3, "dlogits")
    assert dl[:, V:].abs().max() == 0
    am = torch.empty(rows, device=dev, dtype=torch.int64)
    ops.argmax_rows(logits, rows, V, ldv, am)
    assert torch.equal(am, logits[:, :V].float().argmax(-1))


def test_adamw_clip_scaler(ops, dev):
    from neuspeech1_amd.lib import AdamWCfg
    n = 100_003
    p0 = rnd((n,), dev, 1.0, torch.float32, seed=1)
    p = p0.clone()
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: s / 3 if s < 3 else max(0.0, (10 - s) / (10 - 3)))
    cfg = AdamWCfg(1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, 3, 10, 2.0, 0.5, 2000)
    step = torch.zeros(1, device=dev, dtype=torch.int32)
    norm2 = torch.zeros(1, device=dev)
    finf = torch.zeros(1, device=dev, dtype=torch.int32)
    scale = torch.tensor([65536.0], device=dev)
    tracker = torch.zeros(1, device=dev, dtype=torch.int32)
    ws = torch.empty(8192, device=dev, dtype=torch.uint8)
    for it in range(6):
        g_true = rnd((n,), dev, 0.05, torch.float32, seed=10 + it)
        g = g_true * scale  # what the scaled backward leaves in the buffer
        if it == 2:
            g[5] = float("inf")
        ops.grad_norm(g, n, ws, norm2, finf)
        ops.adamw_step(p, g, m, v, n, cfg, step, norm2, finf, scale, tracker)
        if it == 2:
            assert finf.item() == 1 and scale.item() == 32768.0
            continue
        ref_p.grad = g_true.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        sched.step()
    assert step.item() == 5
    close(p, ref_p.detach(), 2e-6, 2e-5, "adamw")


def test_orth_reg_value_and_gradient(ops, dev):
    """ns_orth_reg (AdaLoRA's orthogonality regulariser, peft AdaLoraModel._orthogonal loss: mean over the adapter matrices of
    ||A A^T - I||_F for lora_A (r x in) and ||B^T B - I||_F for lora_B (out x r)): value and gradient against torch autograd,
    live ranks below the padded row stride, lengths that are not a multiple of the staging chunk, both storage orders, the
    loss scale folded into the gradient, accumulation onto a non-zero gradient."""
    torch.manual_seed(0)
    cases = [(12, 512, 0), (12, 2048, 0), (32, 512, 0), (16, 300, 0), (12, 512, 1), (32, 2048, 1), (7, 130, 1), (1, 64, 0)]
    w, scale = 0.5, 1024.0
    mats, grads, refs, jobs = [], [], [], []
    for r, n, is_b in cases:
        rp = 32 if r > 16 else 16
        if is_b:      # (len x r_pad), live columns [0, r)
            Pm = torch.zeros(n, rp, device=dev); Pm[:, :r] = torch.randn(n, r, device=dev) * 0.1
            ld = rp
        else:         # (r_pad x len), live rows [0, r)
            Pm = torch.zeros(rp, n, device=dev); Pm[:r] = torch.randn(r, n, device=dev) * 0.1
            ld = n
        Gm = torch.full_like(Pm, 0.25)
        mats.append(Pm); grads.append(Gm)
        jobs.append((Pm.data_ptr(), Gm.data_ptr(), r, n, ld, is_b))
        q = Pm.clone().requires_grad_(True)
        live = q[:, :r] if is_b else q[:r]
        cov = live.T @ live if is_b else live @ live.T
        refs.append((q, torch.linalg.norm(cov - torch.eye(r, device=dev))))
    table, n = ops.make_orth_jobs(jobs, dev)
    reg = torch.zeros(1, device=dev)
    ls = torch.tensor([scale], device=dev)
    ops.orth_reg(table, n, w / n, ls, reg)
    total = sum(v for _, v in refs) * (w / n)
    close(reg, total.detach().reshape(1), 1e-5, 1e-4, "orth-reg value")
    (total * scale).backward()
    for (q, _), Gm, (r, ln, is_b) in zip(refs, grads, cases):
        close(Gm, 0.25 + q.grad, 1e-3, 1e-3, f"orth-reg gradient r={r} len={ln} is_b={is_b}")


# --------------------------------------------------------------------------- LoRA dropout mask consistency
def _keep_mask(seed, rows, cols, p, dev):
    """numpy restatement of ns_keep_el (csrc/ns_common.h): one hash per (row, col>>2), one byte per element."""
    import numpy as np
    thr8 = int(p * 256.0 + 0.5)
    r = np.arange(rows, dtype=np.uint64)[:, None]
    c4 = (np.arange(cols, dtype=np.uint64) >> 2)[None, :]
    M32 = np.uint64(0xFFFFFFFF)
    x = (np.uint64(seed) ^ ((r * np.uint64(0x9E3779B1)) & M32) ^ ((c4 * np.uint64(0x85EBCA77)) & M32)) & M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    byte = (x >> ((np.arange(cols, dtype=np.uint64) & np.uint64(3)) * np.uint64(8))[None, :]) & np.uint64(0xFF)
    keep = torch.from_numpy((byte >= thr8).astype(np.float32)).to(dev)
    return keep, 256.0 / (256.0 - thr8)


def test_lora_dropout_same_mask_in_forward_dgrad_wgrad(ops, dev):
    M, K, N, r, p, seed = 640, 512, 512, 32, 0.05, 1234567
    x = rnd((M, K), dev, 1.0, seed=1)
    A = rnd((r, K), dev, 0.05, seed=2)
    keep, inv = _keep_mask(seed, M, K, p, dev)
    assert abs(1.0 - keep.mean().item() - 13 / 256) < 5e-3
    xd = (x.float() * keep * inv).half()       # what the kernels see after masking (rounded like they round)
    # forward down-projection: u = drop(x) A^T
    u = torch.empty(M, r, device=dev, dtype=torch.float16)
    # (the kernels apply the mask only; the survivors' 1/keep rides in alpha, here and in the GEMM producing du)
    ops.gemm(A=x, am=ops.rowmap(K), K=K, B=A, ldb=K, M=M, N=r, C16=u, c16m=ops.rowmap(r), flags=ops.NS_GEMM_DROP_A,
             alpha=inv, drop_p=p, drop_seed=seed)
    close(u, xd.float() @ A.float().T, 1e-2, 5e-3, "dropout down-projection")
    # dgrad: dx = dy W + mask * (du A) / (1-p)   (all three GEMM kernels that carry the mask)
    dy, W = rnd((M, N), dev, 0.5, seed=3), rnd((N, K), dev, 0.05, seed=4)
    du0, AT = rnd((M, r), dev, 0.5, seed=5), A.t().contiguous()
    du = (du0.float() * inv).half()           # what the engine's du GEMM emits (alpha = 1/keep)
    ref = dy.float() @ W.float() + keep * (du.float() @ A.float())
    from neuspeech1_amd import lib
    for mode in (0, 2, 3, 4):
        lib.load().ns_debug_set_ring(mode)
        dx = torch.empty(M, K, device=dev, dtype=torch.float16)
        ops.gemm(A=dy, am=ops.rowmap(N), K=N, B=W.t().contiguous(), ldb=N, M=M, N=K, A2=du, am2=ops.rowmap(r), K2=r,
                 B2=AT, ldb2=r, C16=dx, c16m=ops.rowmap(K), drop_p=p, drop_seed=seed)
        close(dx, ref, 2e-2, 5e-3, f"dropout dgrad (ring mode {mode})")
        # wgrad: dA = du^T drop(x)
        dA = torch.zeros(r, K, device=dev)
        ops.gemm(A=du, am=ops.rowmap(r), K=M, B=x, bm=ops.rowmap(K), M=r, N=K, C32=dA, ldc32=K,
                 flags=ops.NS_GEMM_TN | ops.NS_GEMM_ATOMIC32, splits=3, drop_p=p, drop_seed=seed)
        close(dA, du.float().T @ (x.float() * keep), 3e-2, 3e-3, f"dropout wgrad (ring mode {mode})")
    lib.load().ns_debug_set_ring(1)


@pytest.mark.parametrize("M,K,N,p", [(8200, 512, 32, 0.05), (9001, 512, 96, 0.05), (8192, 2048, 32, 0.05), (12345, 512, 32, 0.0),
                                     (96000, 512, 96, 0.05), (8777, 1024, 96, 0.1), (9000, 1280, 32, 0.05), (8200, 768, 96, 0.0)])
def test_streaming_down_projection(ops, dev, M, K, N, p):
    """ns_gemm_skinny (u = alpha * drop(x) A^T at training size: A^T resident in LDS, x streamed in operand layout with a
    permuted reduction index): against torch fp32 with the numpy restatement of the keep mask, ragged row counts, the
    q | k | v width 96, K = 2048 (the fc2 site), a strided destination, and the SAME result as the LDS-staged tile
    kernel it replaces up to the summation order."""
    from neuspeech1_amd import lib
    seed = 424242
    x = rnd((M, K), dev, 1.0, seed=1)
    A = rnd((N, K), dev, 0.05, seed=2)
    keep, inv = _keep_mask(seed, M, K, p, dev) if p > 0 else (torch.ones(M, K, device=dev), 1.0)
    ldu = N + 8
    outs = []
    for mode in (1, 7):
        lib.load().ns_debug_set_ring(mode)
        u = torch.full((M, ldu), float("nan"), device=dev, dtype=torch.float16)
        ops.gemm(A=x, am=ops.rowmap(K), K=K, B=A, ldb=K, M=M, N=N, C16=u, c16m=ops.rowmap(ldu),
                 flags=ops.NS_GEMM_DROP_A if p > 0 else 0, alpha=inv, drop_p=p, drop_seed=seed)
        assert torch.isnan(u[:, N:].float()).all() and not torch.isnan(u[:, :N].float()).any()
        outs.append(u[:, :N].float())
    lib.load().ns_debug_set_ring(1)
    ref = (x.float() * keep * inv) @ A.float().T
    close(outs[0], ref, 1e-2 * (K / 512) ** 0.5, 5e-3, "streaming down-projection")
    close(outs[0], outs[1], 4e-3 * (K / 512) ** 0.5, 2e-3, "streaming vs tile kernel")


@pytest.mark.parametrize("nq,Lk", [(1, 1500), (1, 7), (5, 1500), (3, 130), (8, 33)])
def test_attn_decode_cross_layout(ops, dev, nq, Lk):
    """ns_attn_decode on a per-group K/V block (cross-attention layout): nq query rows of a group share its Lk keys;
    against fp32 softmax(QK^T)V (the kernel takes pre-scaled queries)."""
    groups, H, d = 6, 4, 64
    Q = rnd((groups * nq, H * d), dev, 0.6, seed=1)
    KV = rnd((groups * Lk, 2 * H * d), dev, 0.7, seed=2)
    O = torch.full((groups * nq, H * d), float("nan"), device=dev, dtype=torch.float16)
    ops.attn_decode(Q=Q, K=KV, V=(KV, H * d), O=O, groups=groups, nq=nq, H=H, Lk=Lk, Lk_max=Lk, ldq=H * d, ldk=2 * H * d,
                    ldv=2 * H * d, ldo=H * d, kv_group_stride=Lk)
    q = Q.float().view(groups, nq, H, d).transpose(1, 2)
    k = KV.float()[:, :H * d].view(groups, Lk, H, d).transpose(1, 2)
    v = KV.float()[:, H * d:].view(groups, Lk, H, d).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(groups * nq, H * d)
    close(O, ref, 2e-3, 2e-3, "attn_decode")


@pytest.fixture
def ad_self_form(request):
    from neuspeech1_amd import lib
    lib.load().ns_debug_set_ad_self(request.param)
    yield request.param
    lib.load().ns_debug_set_ad_self(1)


@pytest.mark.parametrize("ad_self_form", [2, 0], indirect=True, ids=["wave_per_head", "four_waves"])
@pytest.mark.parametrize("Lk,H", [(1, 4), (9, 4), (68, 4), (68, 6), (300, 6), (448, 8)])
def test_attn_decode_cache_layout_with_ancestry(ops, dev, Lk, H, ad_self_form):
    """self-attention over the position-major K/V cache: row r at position j reads cache row anc[r][j] (beam ancestry),
    the live length comes from device memory.  Both kernels of the layout: one wave per (row, head) -- the default; heads
    not a multiple of the four waves of a workgroup, key counts over several iterations -- and the four-wave form."""
    rows, d, Lmax = 10, 64, max(80, Lk + 3)
    Q = rnd((rows, 3 * H * d), dev, 0.6, seed=3)
    cache = rnd((Lmax * rows, 2 * H * d), dev, 0.7, seed=4)
    g = torch.Generator(device="cpu").manual_seed(5)
    anc = torch.randint(0, rows, (rows, Lmax), generator=g, dtype=torch.int32).to(dev)
    O = torch.full((rows, H * d), float("nan"), device=dev, dtype=torch.float16)
    klen = torch.tensor([Lk], device=dev, dtype=torch.int32)
    ops.attn_decode(Q=Q, K=cache, V=(cache, H * d), O=O, groups=rows, nq=1, H=H, Lk=Lmax, Lk_max=Lmax, ldq=3 * H * d,
                    ldk=2 * H * d, ldv=2 * H * d, ldo=H * d, anc=anc, anc_ld=Lmax, kv_pos_stride=rows, kv_len_dev=klen)
    c = cache.float().view(Lmax, rows, 2, H, d)
    idx = anc[:, :Lk].long()                                               # (rows, Lk)
    kv = c[torch.arange(Lk, device=dev).unsqueeze(0), idx]                 # (rows, Lk, 2, H, d)
    q = Q.float()[:, :H * d].view(rows, H, 1, d)
    k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).reshape(rows, H * d)
    close(O, ref, 2e-3, 2e-3, "attn_decode anc")
    # append mode: the newest position comes from the projection rows (k | v inside Q's rows) and lands in the cache
    anc2 = anc.clone()
    anc2[:, Lk - 1] = torch.arange(rows, device=dev, dtype=torch.int32)       # a row's newest position is its own slot
    cache2 = cache.clone()
    cache2.view(Lmax, rows, 2 * H * d)[Lk - 1] = 777.0                         # stale: must not be read, must be replaced
    O2 = torch.full_like(O, float("nan"))
    ops.attn_decode(Q=Q, K=cache2, V=(cache2, H * d), O=O2, groups=rows, nq=1, H=H, Lk=Lmax, Lk_max=Lmax, ldq=3 * H * d,
                    ldk=2 * H * d, ldv=2 * H * d, ldo=H * d, anc=anc2, anc_ld=Lmax, kv_pos_stride=rows, kv_len_dev=klen,
                    Knew=(Q, H * d), Vnew=(Q, 2 * H * d), ldnew=3 * H * d)
    want = cache.clone()
    want.view(Lmax, rows, 2 * H * d)[Lk - 1] = Q[:, H * d:]
    assert torch.equal(cache2, want)
    c = want.float().view(Lmax, rows, 2, H, d)
    kv = c[torch.arange(Lk, device=dev).unsqueeze(0), anc2[:, :Lk].long()]
    k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)
    ref2 = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).reshape(rows, H * d)
    close(O2, ref2, 2e-3, 2e-3, "attn_decode append")
    # a ROW RANGE of the batch as its own launch (NS_DECODE_SPLIT's chains): Q / anc / O offset by the range, the cache at its base,
    # slot0 = the range's first slot -- the rows of the range, and only those, are appended and attended
    r0, r1 = 3, 8
    cache3 = cache.clone()
    cache3.view(Lmax, rows, 2 * H * d)[Lk - 1] = 777.0
    O3 = torch.full_like(O, float("nan"))
    ops.attn_decode(Q=Q[r0:r1], K=cache3, V=(cache3, H * d), O=O3[r0:r1], groups=r1 - r0, nq=1, H=H, Lk=Lmax, Lk_max=Lmax, ldq=3 * H * d,
                    ldk=2 * H * d, ldv=2 * H * d, ldo=H * d, anc=anc2[r0:r1], anc_ld=Lmax, kv_pos_stride=rows, kv_len_dev=klen,
                    Knew=(Q[r0:r1], H * d), Vnew=(Q[r0:r1], 2 * H * d), ldnew=3 * H * d, slot0=r0)
    got = cache3.view(Lmax, rows, 2 * H * d)[Lk - 1]
    assert torch.equal(got[r0:r1], Q[r0:r1, H * d:]) and bool((got[:r0] == 777.0).all()) and bool((got[r1:] == 777.0).all())
    if Lk == 1 or bool((anc2[r0:r1, :Lk - 1] >= 0).all()):
        # (older positions may point at slots outside the range: they read the cache as it is -- equal to `want` there)
        assert torch.equal(O3[r0:r1], O2[r0:r1])
    assert bool(torch.isnan(O3[:r0]).all()) and bool(torch.isnan(O3[r1:]).all())


@pytest.mark.parametrize("nq,Lk", [(5, 1500), (1, 1500), (16, 97), (3, 31), (8, 128), (2, 1)])
def test_attn_fewq_and_vt_pack(ops, dev, nq, Lk):
    """ns_vt_pack + ns_attn_fewq (few query rows against a long per-group K/V, MFMA, transposed value image) against
    fp32 softmax(QK^T)V and against ns_attn_decode on the same operands."""
    groups, H, d = 5, 4, 64
    Q = rnd((groups * nq, H * d), dev, 0.6, seed=1)
    KV = rnd((groups * Lk, 2 * H * d), dev, 0.7, seed=2)
    ldvt = (Lk + 31) // 32 * 32
    Vt = torch.full((groups, H, d, ldvt), float("nan"), device=dev, dtype=torch.float16)
    ops.vt_pack((KV, H * d), 2 * H * d, Vt, groups, H, Lk, ldvt)
    v = KV[:, H * d:].view(groups, Lk, H, d)
    assert torch.equal(Vt[..., :Lk], v.permute(0, 2, 3, 1)) and (Vt[..., Lk:] == 0).all()
    O = torch.full((groups * nq, H * d), float("nan"), device=dev, dtype=torch.float16)
    ops.attn_fewq(Q=Q, K=KV, Vt=Vt, O=O, groups=groups, nq=nq, H=H, Lk=Lk, ldq=H * d, ldk=2 * H * d, ldvt=ldvt, ldo=H * d)
    q = Q.float().view(groups, nq, H, d).transpose(1, 2)
    k = KV.float()[:, :H * d].view(groups, Lk, H, d).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v.float().transpose(1, 2)).transpose(1, 2).reshape(groups * nq, H * d)
    close(O, ref, 3e-3, 3e-3, "attn_fewq")
    if nq <= 8:
        O2 = torch.empty_like(O)
        ops.attn_decode(Q=Q, K=KV, V=(KV, H * d), O=O2, groups=groups, nq=nq, H=H, Lk=Lk, Lk_max=Lk, ldq=H * d,
                        ldk=2 * H * d, ldv=2 * H * d, ldo=H * d, kv_group_stride=Lk)
        close(O, O2.float(), 3e-3, 3e-3, "attn_fewq vs attn_decode")


@pytest.mark.parametrize("mode", [2, 3, 4])
def test_wide_nt_kernels_agree_on_every_epilogue_and_ragged_shape(ops, dev, mode):
    """The dispatch picks the 128^2 ring (2), the one-barrier 256^2 ring (3) or the phase-interleaved 256^2 kernel (4) by
    size, so small parity cases never reach the big ones: force each on ragged shapes (M, N, K not multiples of the
    tiles) with every epilogue and operand feature of the descriptor, against the register-staged kernel (mode 0) and
    torch."""
    from neuspeech1_amd import lib
    M, N, K, S, r = 520, 760, 144, 130, 32
    A, B = rnd((M, K), dev, seed=1), rnd((N, K), dev, 0.1, seed=2)
    bias = rnd((N,), dev, 0.2, torch.float32, seed=3)
    R = rnd((M, N), dev, 1.0, torch.float32, seed=4)
    pos = rnd((S, N), dev, 1.0, torch.float32, seed=5)
    P = rnd((M, N), dev, 1.0, seed=6)
    u, Bs = rnd((M, 2 * r), dev, 0.5, seed=7), rnd((N, r), dev, 0.1, seed=8)
    # conv-style operand: 4 segments of 132 output rows, stride-2 overlapping rows over a halo-padded token-major image
    Cin, T2 = 48, 132
    Rc = rnd((4 * T2, N), dev, 1.0, torch.float32, seed=11)
    img = rnd((4, 2 * T2 + 2, Cin), dev, 1.0, seed=9)
    Wc = rnd((N, 3 * Cin), dev, 0.1, seed=10)

    def run():
        out = {}
        C16 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16); G16 = torch.full_like(C16, float("nan"))
        H = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, C16=C16, c16m=ops.rowmap(N), G16=G16,
                 g16m=ops.rowmap(N), R32=R, H32=H, h32m=ops.rowmap(N), pos=pos, pos_rows=S,
                 flags=ops.NS_GEMM_GELU | ops.NS_GEMM_GELU_SAVE_GRAD)
        out["gelu_save"] = (C16, G16, H)
        D = torch.full_like(C16, float("nan"))
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=D, c16m=ops.rowmap(N), P16=P, p16m=ops.rowmap(N),
                 flags=ops.NS_GEMM_MUL_P16, alpha=0.5)
        out["mul_p16"] = (D,)
        D2 = torch.full_like(C16, float("nan"))
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=D2, c16m=ops.rowmap(N), P16=P, p16m=ops.rowmap(N),
                 flags=ops.NS_GEMM_DGELU)
        out["dgelu"] = (D2,)
        E = torch.full_like(C16, float("nan"))
        # second product with two column groups of 384 (a2_ngroup must be a multiple of 128; the last group is ragged)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, A2=u, am2=ops.rowmap(2 * r), K2=r, B2=Bs, ldb2=r,
                 a2_ngroup=384, bias=bias, C16=E, c16m=ops.rowmap(N))
        out["second"] = (E,)
        Hc = Rc.clone()
        ops.gemm(A=(img, 0), am=ops.rowmap(2 * Cin, T2, (2 * T2 + 2) * Cin), K=3 * Cin, B=Wc, ldb=3 * Cin, M=4 * T2, N=N,
                 bias=bias, R32=Hc, H32=Hc, h32m=ops.rowmap(N))
        out["conv_res_inplace"] = (Hc,)
        return out
    try:
        lib.load().ns_debug_set_ring(0)
        ref = run()
        lib.load().ns_debug_set_ring(mode)
        got = run()
    finally:
        lib.load().ns_debug_set_ring(1)
    for k in ref:
        for a, b in zip(got[k], ref[k]):
            assert not torch.isnan(a.float()).any(), k
            close(a, b.float(), 4e-3, 3e-3, f"{k} (mode {mode} vs staged)")
    # and the staged kernel itself against torch on the two least obvious ones
    full = A.float() @ B.float().T
    sec = full + bias
    sec[:, :384] += u[:, :r].float() @ Bs[:384].float().T
    sec[:, 384:] += u[:, r:].float() @ Bs[384:].float().T
    close(ref["second"][0], sec, 6e-3, 3e-3, "second product groups")
    conv = torch.nn.functional.conv1d(img.float().transpose(1, 2), Wc.float().view(N, 3, Cin).permute(0, 2, 1), stride=2)  # (4, N, T2)
    close(ref["conv_res_inplace"][0], Rc + (conv.transpose(1, 2).reshape(4 * T2, N) + bias).half().float(),
          2e-2, 3e-3, "conv rowmap")


@pytest.mark.parametrize("name,N,K,kind", [("qkv_lora", 1536, 512, "second"), ("fc1_gelu", 2048, 512, "gelu"),
                                           ("fc2_residual", 512, 2048, "res"), ("dgrad_mul", 512, 1536, "mulp"),
                                           ("lv2_width", 1280, 1280, "res")])
def test_dominant_gemm_at_bench_size_against_torch(ops, dev, name, N, K, kind):
    """The dominant kernel (phase-interleaved 256^2 NT GEMM: M >= 2048 and >= 192 tiles) at the BENCH row count
    M = 96 000 (B = 64 x 1500 encoder rows) under the automatic dispatch, DIRECTLY against torch fp32 on row slices
    (first / last tile rows, tile seams, a ragged tail is covered by the forced-mode test above): q|k|v with the grouped
    LoRA second product, fc1 + GELU (+ saved gelu'), fc2 + fp32 residual stream, a dgrad with the gelu' multiply."""
    M, r = 96000, 32
    A = rnd((M, K), dev, 1.0, seed=1)
    B = rnd((N, K), dev, K ** -0.5, seed=2)
    bias = rnd((N,), dev, 0.2, torch.float32, seed=3)
    rows = torch.cat([torch.arange(0, 300), torch.arange(47990, 48300), torch.arange(M - 333, M)]).to(dev)
    Ar = A[rows].float()
    full = Ar @ B.float().T + bias
    if kind == "second":
        u, Bs = rnd((M, 3 * r), dev, 0.5, seed=7), rnd((N, r), dev, 0.1, seed=8)
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, A2=u, am2=ops.rowmap(3 * r), K2=r, B2=Bs, ldb2=r,
                 a2_ngroup=N // 3, bias=bias, C16=C, c16m=ops.rowmap(N))
        for j in range(3):
            full[:, j * 512:(j + 1) * 512] += u[rows, j * r:(j + 1) * r].float() @ Bs[j * 512:(j + 1) * 512].float().T
        close(C[rows], full, 6e-3, 3e-3, name)
    elif kind == "gelu":
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        Gd = torch.full_like(C, float("nan"))
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, C16=C, c16m=ops.rowmap(N), G16=Gd,
                 g16m=ops.rowmap(N), flags=ops.NS_GEMM_GELU | ops.NS_GEMM_GELU_SAVE_GRAD)
        pre = full.half().float()
        close(Gd[rows], torch.nn.functional.gelu(pre), 4e-3, 3e-3, name + " gelu")
        x = pre.double().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        close(C[rows], x.grad.float(), 4e-3, 3e-3, name + " saved gelu'")
    elif kind == "res":
        R = rnd((M, N), dev, 1.0, torch.float32, seed=4)
        H = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, R32=R, H32=H, h32m=ops.rowmap(N))
        close(H[rows], R[rows] + full.half().float(), 8e-3, 3e-3, name)
        assert not torch.isnan(H).any()
    else:
        P = rnd((M, N), dev, 1.0, seed=6)
        D = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=D, c16m=ops.rowmap(N), P16=P, p16m=ops.rowmap(N),
                 flags=ops.NS_GEMM_MUL_P16)
        close(D[rows], (Ar @ B.float().T).half().float() * P[rows].float(), 8e-3, 3e-3, name)
        assert not torch.isnan(D.float()).any()


@pytest.mark.parametrize("M,N,r,G,splits", [(1000, 512, 32, 1, 0), (520, 256, 16, 3, 3), (777, 2048, 32, 1, 5), (4096, 512, 32, 3, 0),
                                            (333, 1280, 32, 1, 2), (64, 256, 32, 1, 0), (96000, 512, 32, 3, 0)])
@pytest.mark.parametrize("slabs", [True, False])
def test_lora_backward_du_and_dB_in_one_pass(ops, dev, M, N, r, G, splits, slabs):
    """ns_lora_bwd_dudb (du_g = alpha dy_g sB_g and dB_g += alpha_g dy_g^T u_g from ONE pass over dy) against torch fp32,
    ragged row counts, every built (N, G) shape class, the padded rank 16, accumulation into a non-zero dB, and the
    bench size M = 96 000 (q | k | v); dB partials through the workspace slabs + reduce launch, and through fp32 atomics.
    The slab form is bitwise reproducible."""
    assert ops.lora_bwd_supported(N, r, G) and not ops.lora_bwd_supported(N + 8, r, G) and not ops.lora_bwd_supported(N, 24, G)
    dy = rnd((M, G * N), dev, 0.5, seed=1)
    u = rnd((M, G * r), dev, 0.5, seed=2)
    sBT = [rnd((r, N), dev, 0.1, seed=3 + g) for g in range(G)]
    dB0 = [rnd((N, r), dev, 1.0, torch.float32, seed=10 + g) for g in range(G)]
    dB = [t.clone() for t in dB0]
    du = torch.full((M, G * r), float("nan"), device=dev, dtype=torch.float16)
    al = [0.7, 1.3, 2.0][:G]
    ops.lora_bwd_dudb(dy=dy, ldy=G * N, u=u, ldu=G * r, du=du, lddu=G * r, sBT=sBT, dB=dB, lddb=r, M=M, N=N, r=r, alpha_du=1.25,
                      alpha_db=al, splits=splits, slabs=slabs)
    if slabs:
        dB2 = [t.clone() for t in dB0]
        ops.lora_bwd_dudb(dy=dy, ldy=G * N, u=u, ldu=G * r, du=du, lddu=G * r, sBT=sBT, dB=dB2, lddb=r, M=M, N=N, r=r, alpha_du=1.25,
                          alpha_db=al, splits=splits, slabs=True)
        assert all(torch.equal(a, b) for a, b in zip(dB, dB2))
    for g in range(G):
        dyg = dy[:, g * N:(g + 1) * N].float()
        close(du[:, g * r:(g + 1) * r], 1.25 * dyg @ sBT[g].float().T, 2e-2 * (N / 512) ** 0.5, 4e-3, f"du group {g}")
        ref = dB0[g] + al[g] * (dyg.T @ u[:, g * r:(g + 1) * r].float())
        close(dB[g], ref, 2e-3 * max(1.0, (M / 1000) ** 0.5), 2e-3, f"dB group {g}")
    assert not torch.isnan(du.float()).any()


@pytest.mark.parametrize("M,N,K,splits", [(300, 512, 51968, 12), (2816, 512, 51968, 12), (130, 200, 4096, 3)])
def test_gemm_nt_split_k_into_fp32(ops, dev, M, N, K, splits):
    """split-K NT GEMM (the LM-head dgrad: K = the padded vocabulary against a handful of output tiles): fp32 atomics
    into a zeroed C32, against torch fp32; ragged M / N, K ranges that do not divide evenly."""
    A = rnd((M, K), dev, 0.05, seed=1)
    B = rnd((N, K), dev, 0.05, seed=2)
    C = torch.zeros(M, N, device=dev)
    ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C32=C, ldc32=N, splits=splits)
    ref = A.float() @ B.float().T
    close(C, ref, 2e-3, 2e-3, "split-K NT")


@pytest.mark.parametrize("form", ["p8s"])
def test_persistent_gemm_is_bitwise_the_one_tile_form(ops, dev, form):
    """ns_gemm_p8s (one workgroup per CU walks several 256 x 256 tiles; the default at >= 700 tiles) against ns_gemm_p8 (one tile per
    workgroup) on ragged shapes with > 256 tiles, so that workgroups really carry the next tile's prologue through an epilogue:
    every epilogue kind, second product (ragged column groups, LoRA dropout), segmented row maps with an in-place residual, the
    GELU side product.  Same arithmetic in the same order: outputs must be bit-identical."""
    from neuspeech1_amd import lib
    M, N, K, r = 256 * 90 + 40, 760, 208, 32
    A, B = rnd((M, K), dev, seed=1), rnd((N, K), dev, 0.1, seed=2)
    bias = rnd((N,), dev, 0.2, torch.float32, seed=3)
    R = rnd((M, N), dev, 1.0, torch.float32, seed=4)
    P = rnd((M, N), dev, 1.0, seed=6)
    u, Bs = rnd((M, 2 * r), dev, 0.5, seed=7), rnd((N, r), dev, 0.1, seed=8)
    u3, B3 = rnd((M, 3 * r), dev, 0.5, seed=12), rnd((N, 3 * r), dev, 0.1, seed=13)
    Cin, T2, segs = 48, 3000, 8
    Rc = rnd((segs * T2, N), dev, 1.0, torch.float32, seed=11)
    img = rnd((segs, 2 * T2 + 2, Cin), dev, 1.0, seed=9)
    Wc = rnd((N, 3 * Cin), dev, 0.1, seed=10)
    Ns = 768
    Bside, sideB = rnd((Ns, K), dev, 0.1, seed=14), rnd((32, Ns), dev, 0.1, seed=15)
    nan16 = lambda n: torch.full((M, n), float("nan"), device=dev, dtype=torch.float16)

    def run():
        out = {}
        C16, G16 = nan16(N), nan16(N)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, C16=C16, c16m=ops.rowmap(N), G16=G16, g16m=ops.rowmap(N),
                 flags=ops.NS_GEMM_GELU | ops.NS_GEMM_GELU_SAVE_GRAD)
        out["gelu_save"] = (C16, G16)
        H = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, R32=R, H32=H, h32m=ops.rowmap(N),
                 A2=u, am2=ops.rowmap(2 * r), K2=r, B2=Bs, ldb2=r, a2_ngroup=384, drop_p=0.05, drop_seed=5)
        out["res_second_drop"] = (H,)
        D = nan16(N)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=D, c16m=ops.rowmap(N), P16=P, p16m=ops.rowmap(N),
                 flags=ops.NS_GEMM_MUL_P16, alpha=0.5, A2=u3, am2=ops.rowmap(3 * r), K2=3 * r, B2=B3, ldb2=3 * r)
        out["mul_p16_k2_96"] = (D,)
        D2 = nan16(N)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=D2, c16m=ops.rowmap(N), P16=P, p16m=ops.rowmap(N), flags=ops.NS_GEMM_DGELU)
        out["dgelu"] = (D2,)
        E = nan16(N)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, A2=u, am2=ops.rowmap(2 * r), K2=16, B2=Bs, ldb2=r, a2_ngroup=384, C16=E, c16m=ops.rowmap(N))
        out["second_k2_16"] = (E,)
        # the persistent form fetches the second product's first round through LDS when a tile lies in ONE column group (a2_ngroup % 256 == 0):
        # two groups of 512 columns (the second ragged), K2 = 32; no groups, K2 = 16 (half of each 64-B image row comes from the range check)
        E2 = nan16(N)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, A2=u, am2=ops.rowmap(2 * r), K2=r, B2=Bs, ldb2=r, a2_ngroup=512, bias=bias,
                 C16=E2, c16m=ops.rowmap(N), drop_p=0.05, drop_seed=9)
        out["second_groups_512"] = (E2,)
        E3 = nan16(N)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=B, ldb=K, M=M, N=N, A2=u, am2=ops.rowmap(2 * r), K2=16, B2=Bs, ldb2=r, C16=E3, c16m=ops.rowmap(N))
        out["second_k2_16_one_group"] = (E3,)
        Hc = Rc.clone()
        ops.gemm(A=(img, 0), am=ops.rowmap(2 * Cin, T2, (2 * T2 + 2) * Cin), K=3 * Cin, B=Wc, ldb=3 * Cin, M=segs * T2, N=N,
                 bias=bias, R32=Hc, H32=Hc, h32m=ops.rowmap(N))
        out["conv_res_inplace"] = (Hc,)
        Cs, Gs = nan16(Ns), nan16(Ns)
        slab = torch.full((Ns // 256, M, 32), float("nan"), device=dev)
        ops.gemm(A=A, am=ops.rowmap(K), K=K, B=Bside, ldb=K, M=M, N=Ns, bias=bias[:Ns].contiguous(), C16=Cs, c16m=ops.rowmap(Ns), G16=Gs, g16m=ops.rowmap(Ns),
                 flags=ops.NS_GEMM_GELU | ops.NS_GEMM_GELU_SAVE_GRAD, side_B=sideB, side_ldb=Ns, side_n=32, side_out=slab, side_drop_p=0.05,
                 side_drop_seed=3)
        out["gelu_side"] = (Cs, Gs, slab)
        return out
    try:
        lib.load().ns_debug_set_ring(4)
        ref = run()
        lib.load().ns_debug_set_ring(9)
        got = run()
    finally:
        lib.load().ns_debug_set_ring(1)
    for k in ref:
        for a, b in zip(got[k], ref[k]):
            assert not torch.isnan(a.float()).any(), k
            assert torch.equal(a, b), (k, int((a != b).sum()))
    # and the one-tile form against torch where no other test covers the combination (ragged second-product groups under dropout are covered
    # by test_lora_dropout_*; here the plain second product with K2 = 16)
    sec = A.float() @ B.float().T
    sec[:, :384] += u[:, :16].float() @ Bs[:384, :16].float().T
    sec[:, 384:] += u[:, 16:32].float() @ Bs[384:, :16].float().T      # group g reads columns [g K2, (g + 1) K2) of A2
    close(ref["second_k2_16"][0], sec, 6e-3, 3e-3, "second product, K2 = 16")
    sec1 = A.float() @ B.float().T + u[:, :16].float() @ Bs[:, :16].float().T
    close(got["second_k2_16_one_group"][0], sec1, 6e-3, 3e-3, "second product through LDS, K2 = 16")


@pytest.mark.parametrize("M,K,K2", [(96000, 512, 32), (96000, 2048, 32), (4500, 512, 0), (4500, 2048, 16), (1500 * 7, 1024, 32)])
def test_gemm_ln_is_bitwise_the_two_launches(ops, dev, M, K, K2):
    """ns_gemm_ln (csrc/ns_gemm_rowln.hip: residual Linear with N = 512 + the LayerNorm that reads its result, a workgroup owning
    complete rows) against ns_gemm followed by ns_layernorm_fwd on the same operands: H32, x16, mean and rstd BITWISE equal
    (same MFMA products in the same order, the same LayerNorm sums and shuffle tree), at the bench's row count, with a ragged last
    row tile, with and without the LoRA second product (K2 = 32, and 16 = a zero-padded rank), against torch fp32 as well."""
    N = 512
    assert ops.gemm_ln_supported(M, N, K, K2)
    A = rnd((M, K), dev, 1.0, seed=1)
    W = rnd((N, K), dev, K ** -0.5, seed=2)
    bias = rnd((N,), dev, 0.1, torch.float32, seed=3)
    R = rnd((M, N), dev, 2.0, torch.float32, seed=4)
    gamma = rnd((N,), dev, 1.0, torch.float32, seed=5) * 0.2 + 1.0
    beta = rnd((N,), dev, 0.1, torch.float32, seed=6)
    kw = dict(A=A, am=ops.rowmap(K), K=K, B=W, ldb=K, M=M, N=N, bias=bias, R32=R, h32m=ops.rowmap(N))
    if K2:
        u = rnd((M, 32), dev, 0.5, seed=7)
        sB = rnd((N, 32), dev, 0.2, seed=8)
        if K2 == 16:
            u[:, 16:] = 0
        kw.update(A2=u, am2=ops.rowmap(32), K2=K2, B2=sB, ldb2=32)
    H0 = torch.empty(M, N, device=dev)
    x0 = torch.empty(M, N, device=dev, dtype=torch.float16)
    m0, r0 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.gemm(H32=H0, **kw)
    ops.layernorm_fwd(H0, gamma, beta, x0, m0, r0, M, N)
    H1 = torch.full((M, N), float("nan"), device=dev)
    x1 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
    m1, r1 = torch.full((M,), float("nan"), device=dev), torch.full((M,), float("nan"), device=dev)
    ops.gemm_ln(H32=H1, gamma=gamma, beta=beta, x16=x1, ldx=N, mean=m1, rstd=r1, **kw)
    torch.cuda.synchronize()
    if ((M + 255) // 256) * 2 >= 192:
        # ns_gemm takes these shapes to the phase-interleaved 256 x 256 kernel, whose products and their order ns_gemm_ln repeats
        assert torch.equal(H1, H0), f"H32 differs in {(H1 != H0).sum().item()} places, max {(H1 - H0).abs().max().item()}"
        assert torch.equal(m1, m0) and torch.equal(r1, r0)
        assert torch.equal(x1.view(torch.int16), x0.view(torch.int16))
    else:
        # fewer than 192 tiles: ns_gemm runs its 128 x 128 ring kernel (v_mfma_f32_32x32x16_f16: another summation order), so the
        # Linear's fp16 output may differ by one rounding here and there
        close(H1, H0, 4e-3, 1e-3, "H32 vs the two launches")
        close(x1, x0, 1e-2, 1e-2, "x16 vs the two launches")
        close(m1, m0, 1e-4, 1e-4, "mean")
        close(r1, r0, 1e-4, 1e-3, "rstd")
    # and against torch fp32 on a row slice (the two-launch path has its own tests; this guards the comparison itself)
    sl = slice(M - 300, M)
    y = A[sl].float() @ W.float().t() + bias
    if K2:
        y = y + u[sl, :K2].float() @ sB[:, :K2].float().t()
    h = R[sl] + y.half().float()
    close(H1[sl], h, 2e-2, 2e-3, "H32 vs torch")
    close(x1[sl], F.layer_norm(h, (N,), gamma, beta, 1e-5), 2e-2, 1e-2, "x16 vs torch")


def test_zero_spans_and_counter_kernels(ops, dev):
    """ns_zero_spans (what clears the gradient buffer / split-K targets of a captured step and every engine allocation) and ns_add_i32
    (the LoRA-dropout step counter): 16-B body + dword tail, more than 8 spans per call, neighbours untouched, empty tensors skipped."""
    sizes = [1, 3, 4, 5, 1023, 4096, 14_650_000 // 4, 7, 64, 2, 9, 33]
    bufs = [torch.full((n + 8,), 7.0, device=dev) for n in sizes]
    views = [b[4:4 + n] for b, n in zip(bufs, sizes)]          # (a 16-byte aligned start: the ABI's contract)
    ops.zero_(*views, None, torch.empty(0, device=dev))
    torch.cuda.synchronize()
    for b, n in zip(bufs, sizes):
        assert float(b[4:4 + n].abs().max()) == 0.0, n
        assert float(b[:4].min()) == 7.0 and float(b[4 + n:].min()) == 7.0, n
    h = torch.full((5, 3), 1.0, device=dev, dtype=torch.float16)          # 30 bytes is not a multiple of 4: refused loudly, not half-cleared
    from neuspeech1_amd.lib import NeuSpeechHipError
    with pytest.raises(NeuSpeechHipError, match="ns_zero_spans"):
        ops.zero_(h)
    c = torch.tensor([5, 40], device=dev, dtype=torch.int32)
    ops.add_i32(c, 1)
    ops.add_i32((c, 1), -3)
    assert c.tolist() == [6, 37]
    ops.add_i32(c, 2, n=2)                  # the decode loop's position / length pair: one launch
    assert c.tolist() == [8, 39]
    with pytest.raises(NeuSpeechHipError, match="ns_add_i32"):
        ops.add_i32(c, 1, n=65)
    z = ops.zeros(3, 5, device=dev, dtype=torch.int64)
    assert z.shape == (3, 5) and int(z.abs().sum()) == 0


def test_gemm_ln_refuses_what_it_does_not_build(ops, dev):
    """ns_gemm_ln: N != 512, K % 64 != 0, extra outputs or flags come back as errors (the engine then takes the two launches)."""
    from neuspeech1_amd.lib import NeuSpeechHipError
    assert not ops.gemm_ln_supported(96000, 1280, 1280, 32) and not ops.gemm_ln_supported(96000, 512, 544, 0)
    assert not ops.gemm_ln_supported(512, 512, 512, 0) and not ops.gemm_ln_supported(96000, 512, 512, 96)
    M, N, K = 2048, 512, 512
    A, W = rnd((M, K), dev), rnd((N, K), dev, 0.05)
    R, H = rnd((M, N), dev, 1.0, torch.float32), torch.empty(M, N, device=dev)
    g, b_ = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    x = torch.empty(M, N, device=dev, dtype=torch.float16)
    base = dict(A=A, am=ops.rowmap(K), K=K, B=W, ldb=K, M=M, N=N, R32=R, H32=H, h32m=ops.rowmap(N), gamma=g, beta=b_, x16=x, ldx=N)
    ops.gemm_ln(**base)                                             # the plain call works
    for bad in (dict(flags=ops.NS_GEMM_GELU), dict(C16=x, c16m=ops.rowmap(N)), dict(drop_p=0.05), dict(K=K - 32)):
        with pytest.raises(NeuSpeechHipError, match="ns_gemm_ln"):
            ops.gemm_ln(**{**base, **bad})
    torch.cuda.synchronize()
    ref = R + (A.float() @ W.float().t()).half().float()
    close(H, ref, 2e-2, 2e-3, "H32")
    close(x, F.layer_norm(ref, (N,), g, b_, 1e-5), 2e-2, 1e-2, "x16")

