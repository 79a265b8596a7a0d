"""Where a p8 tile spends its cycles: per-block s_memtime stamps (library built with NS_EXTRA_HIPCC_FLAGS=-DNS_P8_STAMPS)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap
dev = torch.device("cuda:0")
lib.load().ns_debug_set_ring(4)
import os
CASES = [(96000, 1536, 512, "c16"), (96000, 512, 512, "res"), (96000, 2048, 512, "gelu"), (96000, 1536, 64, "c16")]
if os.environ.get("CASES") == "occ":
    CASES = [(256 * 32, 256, 512, "c16"), (256 * 128, 256, 512, "c16"), (256 * 256, 256, 512, "c16"), (256 * 32, 256, 512, "res"), (256 * 256, 256, 512, "res")]
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
        kw.update(bias=torch.randn(N, device=dev), C16=C, c16m=rowmap(N), G16=G, g16m=rowmap(N), flags=1 | (1 << 27))
    for _ in range(3):
        ops.gemm(**kw)
    torch.cuda.synchronize()
    s = st.view(tiles, 16).cpu().double()
    d = lambda a, b: (s[:, b] - s[:, a])
    names = ["setup", "prologue(issue+wait)", "mainloop", "prefetch+stage+barrier", "settle loads", "finish (stores)", "-"]
    print(f"--- {epi} M={M} N={N} K={K}: tiles={tiles}")
    for i, nm in enumerate(names):
        v = d(i, i + 1)
        print(f"  {nm:24s} mean {v.mean():9.0f} cyc   p10 {v.quantile(0.1):9.0f}  p90 {v.quantile(0.9):9.0f}")
    tot = d(0, 7)
    print(f"  {'total in-kernel':24s} mean {tot.mean():9.0f} cyc")
    rt = (s[:, 9] - s[:, 8]) * 10.0   # 100 MHz ticks -> ns
    print(f"  wall per block {rt.mean():.0f} ns  => clock ~{tot.mean() / rt.mean():.2f} GHz;  launch span {(s[:, 9].max() - s[:, 8].min()) * 10 / 1000:.1f} us;"
          f" sum of block walls / 256 CUs = {rt.sum() / 256 / 1000:.1f} us")
