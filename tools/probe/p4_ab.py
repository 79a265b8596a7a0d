"""ns_gemm_p4 (128 x 256 tiles, two workgroups per CU) against ns_gemm_p8 / p8s at the training step's large-M shapes (M = 96 000),
per launch class, operands rotated over NSET buffer sets (cold, as in the step).  Prints us per launch for both and the ratio."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuspeech1_amd import lib, ops  # noqa: E402
from neuspeech1_amd.ops import NS_GEMM_GELU, NS_GEMM_GELU_SAVE_GRAD, NS_GEMM_MUL_P16, rowmap  # noqa: E402

dev = torch.device("cuda:0")
M, NSET, REP = int(os.environ.get("M", 96000)), 2, 5
so = lib.load()
seed_dev = torch.zeros(1, device=dev, dtype=torch.int32)


def bufs(K, N, s):
    g = torch.Generator(device=dev).manual_seed(s)
    r = lambda *sh, sc=1.0, dt=torch.float16: (torch.randn(*sh, device=dev, generator=g) * sc).to(dt)  # noqa: E731
    return dict(A=r(M, K), W=r(N, K, sc=K ** -0.5), bias=r(N, sc=0.1, dt=torch.float32), u=r(M, 96, sc=0.5), sB=r(N, 96, sc=0.2),
                C=torch.empty(M, N, device=dev, dtype=torch.float16), G=torch.empty(M, N, device=dev, dtype=torch.float16),
                P=r(M, N), R=r(M, N, dt=torch.float32) if N <= 512 else None, H=torch.empty(M, N, device=dev) if N <= 512 else None,
                sideB=r(32, N, sc=0.1), slab=torch.empty((N // 256) * M * 32, device=dev))


CASES = {
    # name: (K, N, p4 class bit, kwargs builder)
    "qkv fwd (plain + K2=32 groups)": (512, 1536, 1, lambda d: dict(C16=d["C"], c16m=rowmap(1536), A2=d["u"], am2=rowmap(96), K2=32, B2=d["sB"], ldb2=32, a2_ngroup=512)),
    "ckv fwd (plain)": (512, 1024, 1, lambda d: dict(C16=d["C"], c16m=rowmap(1024))),
    "fc1 fwd (GELU + save + side + K2)": (512, 2048, 4, lambda d: dict(C16=d["C"], c16m=rowmap(2048), G16=d["G"], g16m=rowmap(2048), flags=NS_GEMM_GELU | NS_GEMM_GELU_SAVE_GRAD,
                                                                      A2=d["u"], am2=rowmap(96), K2=32, B2=d["sB"], ldb2=32, side_B=d["sideB"], side_ldb=2048, side_n=32,
                                                                      side_out=d["slab"], side_drop_p=0.05, side_drop_seed=3, seed_dev=seed_dev)),
    "fc1 fwd (GELU + save, no side)": (512, 2048, 2, lambda d: dict(C16=d["C"], c16m=rowmap(2048), G16=d["G"], g16m=rowmap(2048), flags=NS_GEMM_GELU | NS_GEMM_GELU_SAVE_GRAD)),
    "fc2 dgrad (x P16 + drop K2)": (512, 2048, 32, lambda d: dict(C16=d["C"], c16m=rowmap(2048), P16=d["P"], p16m=rowmap(2048), flags=NS_GEMM_MUL_P16, A2=d["u"], am2=rowmap(96),
                                                                K2=32, B2=d["sB"], ldb2=32, drop_p=0.05, drop_seed=7, seed_dev=seed_dev)),
    "out fwd (res + K2)": (512, 512, 8, lambda d: dict(R32=d["R"], H32=d["H"], h32m=rowmap(512), A2=d["u"], am2=rowmap(96), K2=32, B2=d["sB"], ldb2=32)),
    "fc2 fwd (res + K2), K=2048": (2048, 512, 8, lambda d: dict(R32=d["R"], H32=d["H"], h32m=rowmap(512), A2=d["u"], am2=rowmap(96), K2=32, B2=d["sB"], ldb2=32)),
    "fc1 dgrad (drop K2), K=2048": (2048, 512, 64, lambda d: dict(C16=d["C"], c16m=rowmap(512), A2=d["u"], am2=rowmap(96), K2=32, B2=d["sB"], ldb2=32, drop_p=0.05, drop_seed=7, seed_dev=seed_dev)),
    "qkv dgrad (drop K2=96), K=1536": (1536, 512, 64, lambda d: dict(C16=d["C"], c16m=rowmap(512), A2=d["u"], am2=rowmap(96), K2=96, B2=d["sB"], ldb2=96, drop_p=0.05, drop_seed=7, seed_dev=seed_dev)),
    "out dgrad (drop K2), K=512": (512, 512, 64, lambda d: dict(C16=d["C"], c16m=rowmap(512), A2=d["u"], am2=rowmap(96), K2=32, B2=d["sB"], ldb2=32, drop_p=0.05, drop_seed=7, seed_dev=seed_dev)),
    "ckv dgrad all (res), K=6144": (6144, 512, 8, lambda d: dict(H32=d["H"], h32m=rowmap(512))),
}
only = os.environ.get("ONLY")
for name, (K, N, bit, mk) in CASES.items():
    if only and only not in name:
        continue
    sets = [bufs(K, N, s) for s in range(NSET)]
    res = {}
    for form, mask in (("p8", 0), ("p4", 127)):
        so.ns_debug_set_p4(mask)
        def fn(d):
            ops.gemm(A=d["A"], am=rowmap(K), K=K, B=d["W"], ldb=K, M=M, N=N, bias=d["bias"], **mk(d))
        for d in sets:
            fn(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            for d in sets:
                fn(d)
        e1.record()
        torch.cuda.synchronize()
        res[form] = e0.elapsed_time(e1) / (REP * NSET) * 1e3
    so.ns_debug_set_p4(-1)
    fl = 2.0 * M * N * K
    print(f"{name:36s} K={K:5d} N={N:5d}: p8 {res['p8']:7.1f} us ({fl / res['p8'] / 1e6:5.0f} TF)   p4 {res['p4']:7.1f} us ({fl / res['p4'] / 1e6:5.0f} TF)   "
          f"p4/p8 {res['p4'] / res['p8']:.3f}", flush=True)
    del sets
    torch.cuda.empty_cache()
