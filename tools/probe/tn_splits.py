"""Split count scan for the LoRA weight-gradient TN GEMMs (dB = dy^T u: N x r, dA = du^T x: r x K)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops
from neuspeech1_amd.ops import rowmap, NS_GEMM_TN, NS_GEMM_ATOMIC32
dev = torch.device("cuda:0")
M = 96000
rnd = lambda *s: (torch.randn(*s, device=dev)).half()
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for name, No, Ko, drop in (("dB N=512", 512, 32, 0.0), ("dB N=2048", 2048, 32, 0.0), ("dB N=1536(q|k|v: 3 x 512)", 512, 32, 0.0),
                           ("dA K=512 drop", 32, 512, 0.05), ("dA q|k|v (3r=96) K=512 drop", 96, 512, 0.05), ("dA K=2048 drop", 32, 2048, 0.05), ("dA K=512", 32, 512, 0.0)):
    A, B = rnd(M, No), rnd(M, Ko)
    C = torch.zeros(No, Ko, device=dev)
    row = []
    for sp in (24, 48, 96, 192, 384, 768):
        f = lambda: ops.gemm(A=A, am=rowmap(No), K=M, B=B, bm=rowmap(Ko), M=No, N=Ko, C32=C, ldc32=Ko,
                             flags=NS_GEMM_TN | NS_GEMM_ATOMIC32, splits=sp, drop_p=drop, drop_seed=3)
        row.append((sp, t(f)))
    mb = M * (No + Ko) * 2 / 1e6
    print(f"{name:28s} " + "  ".join(f"s{sp}: {ms*1000:6.1f}us" for sp, ms in row) + f"   ({mb:.0f} MB -> {mb/1e3/min(ms for _, ms in row):.2f} TB/s best)", flush=True)
