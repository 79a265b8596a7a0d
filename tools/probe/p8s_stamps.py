"""Where a tile of the persistent kernel (mode 9) spends its cycles (library built with NS_EXTRA_HIPCC_FLAGS=-DNS_P8_STAMPS)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap
dev = torch.device("cuda:0")
CASES = [(96000, 2048, 512, "gelu"), (96000, 2048, 512, "gelu_side")]
MODES = (9,)
for mode in MODES:
  lib.load().ns_debug_set_ring(mode)
  for M, N, K, epi in CASES:
    A = torch.randn(M, K, device=dev).half(); B = (torch.randn(N, K, device=dev) * 0.02).half()
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    st = torch.zeros(tiles * 16, device=dev, dtype=torch.int64)
    kw = dict(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C32=st.view(torch.float32), ldc32=N)
    if epi == "c16":
        kw.update(C16=C, c16m=rowmap(N), flags=1 << 27)
    elif epi == "res":
        R = torch.randn(M, N, device=dev); H = torch.empty_like(R)
        kw.update(bias=torch.randn(N, device=dev), R32=R, H32=H, h32m=rowmap(N), flags=1 << 27)
    else:
        G = torch.empty_like(C)
        kw.update(bias=torch.randn(N, device=dev), C16=C, c16m=rowmap(N), G16=G, g16m=rowmap(N), flags=1 | 32 | (1 << 27))
        if epi == "gelu_nograd":
            kw.update(flags=1 | (1 << 27), C16=None, c16m=None)
        if epi == "gelu_side":
            SB = (torch.randn(32, N, device=dev) * 0.05).half()
            slabs = torch.empty((N // 256) * M * 32, device=dev)
            kw.update(side_B=SB, side_ldb=N, side_n=32, side_out=slabs, side_drop_p=0.05, side_drop_seed=7)
    for _ in range(3):
        ops.gemm(**kw)
    torch.cuda.synchronize()
    s = st.view(tiles, 16).cpu().double()
    d = lambda a, b: (s[:, b] - s[:, a])
    if mode == 4:
        names = ["setup", "prologue(issue+wait)", "mainloop", "prefetch+stage+barrier", "settle loads", "finish (stores)", "-"]
        last = 7
    else:
        names = ["k2+wait prologue", "mainloop", "stage0+dma issue", "finish half 0", "stage half 1", "finish half 1"]
        last = 6
    print(f"--- mode {mode} {epi} M={M} N={N} K={K}: tiles={tiles}")
    for i, nm in enumerate(names):
        v = d(i, i + 1)
        print(f"  {nm:24s} mean {v.mean():9.0f} cyc   p10 {v.quantile(0.1):9.0f}  p90 {v.quantile(0.9):9.0f}")
    if mode == 9:
        for a, b, nm in ((2, 7, "  .. bias read + next prologue issue"), (7, 10, "  .. stage half 0 (LDS writes)"), (10, 3, "  .. barrier")):
            v = d(a, b); print(f"  {nm:40s} mean {v.mean():9.0f} cyc   p10 {v.quantile(0.1):9.0f}  p90 {v.quantile(0.9):9.0f}")
    tot = d(0, last)
    print(f"  {'total in-tile':24s} mean {tot.mean():9.0f} cyc;  span of all stamps {(s[:, :last+1].max() - s[:, 0].min()):.0f} cyc")
