"""GEMM ablation on the GPU box: full / no-loads / no-compute / neither (debug flag bits 30 / 29) per kernel variant."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops, lib
from neuspeech1_amd.ops import rowmap
dev = torch.device("cuda:0")
modes = [int(m) for m in os.environ.get("MODES", "3,4").split(",")]
shapes = [(96000, 1536, 512), (96000, 512, 2048), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    A = (torch.randn(M, K, device=dev)).half(); B = (torch.randn(N, K, device=dev) * 0.02).half()
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    def run(flags, name):
        f = lambda: ops.gemm(A=A, am=rowmap(K), K=K, B=B, ldb=K, M=M, N=N, C16=C, c16m=rowmap(N), flags=flags)
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"  {name:28s} {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.0f} TF/s-equiv", flush=True)
    for mode in modes:
        lib.load().ns_debug_set_ring(mode)
        print(f"shape {M}x{N}x{K} ring mode {mode}")
        run(0, "full")
        run(1 << 30, "no loads (mfma+lds only)")
        run(1 << 29, "no compute (loads only)")
        run((1 << 29) | (1 << 30), "neither (barriers+epilogue)")
