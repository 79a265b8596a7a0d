"""Per-tile fixed cost vs K-loop cost: time(K) for K = 64..2048 at M=96000 (C16-only epilogue and fp32-residual epilogue)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap
dev = torch.device("cuda:0")
modes = [int(m) for m in os.environ.get("MODES", "3,4").split(",")]
XF = int(os.environ.get("XFLAGS", "0"))
M = 96000
def t(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
for N, epi in ((1536, "c16"), (512, "res"), (2048, "gelu")):
    for mode in modes:
        lib.load().ns_debug_set_ring(mode)
        row = []
        for K in (64, 128, 256, 512, 1024, 2048):
            A = torch.randn(M, K, device=dev).half(); B = (torch.randn(N, K, device=dev) * 0.02).half()
            C = torch.empty(M, N, device=dev, dtype=torch.float16)
            if epi == "c16":
                f = lambda: ops.gemm(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=C, c16m=rowmap(N), flags=XF)
            elif epi == "res":
                R = torch.randn(M, N, device=dev); H = torch.empty_like(R); bias = torch.randn(N, device=dev)
                f = lambda: ops.gemm(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, R32=R, H32=H, h32m=rowmap(N), flags=XF)
            else:
                G = torch.empty_like(C); bias = torch.randn(N, device=dev)
                f = lambda: ops.gemm(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, bias=bias, C16=C, c16m=rowmap(N), G16=G, g16m=rowmap(N), flags=1 | XF)
            row.append(t(f))
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        rounds = tiles / 256.0
        print(f"N={N} {epi} mode {mode}: " + " ".join(f"K{k}={v*1000:.0f}us" for k, v in zip((64, 128, 256, 512, 1024, 2048), row)) +
              f" | per-tile-round fixed ~{row[0]*1000/rounds:.1f}us, per 64-slice ~{(row[5]-row[4])*1000/16/rounds:.2f}us", flush=True)
