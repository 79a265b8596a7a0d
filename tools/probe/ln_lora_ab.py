import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from neuspeech1_amd import ops
dev = torch.device("cuda:0")
M, d = 96000, 512
x = torch.randn(M, d, device=dev); g = torch.randn(d, device=dev); b = torch.randn(d, device=dev)
y = torch.empty(M, d, device=dev, dtype=torch.float16); m = torch.empty(M, device=dev); r = torch.empty(M, device=dev)
big = torch.empty(600 * 1024 * 1024, device=dev, dtype=torch.uint8)
def t(fn, n=20):
    best = 1e9
    for rep in range(3):
        tot = 0
        for _ in range(n):
            big.zero_()   # flush the Infinity Cache between calls (cold operands, as in the step)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        best = min(best, tot / n)
    return best * 1e3
ctr = torch.zeros(1, device=dev, dtype=torch.int32)
for n_out in (32, 96):
    A = (torch.randn(n_out, d, device=dev) * 0.05).half()
    u = torch.empty(M, n_out, device=dev, dtype=torch.float16)
    t_ln = t(lambda: ops.layernorm_fwd(x, g, b, y, m, r, M, d))
    t_sk = t(lambda: ops.gemm(A=y, am=ops.rowmap(d), K=d, B=A, ldb=d, M=M, N=n_out, C16=u, c16m=ops.rowmap(n_out), flags=ops.NS_GEMM_DROP_A, alpha=1.05, drop_p=0.05, drop_seed=1, seed_dev=ctr))
    t_f = t(lambda: ops.layernorm_fwd_lora(x, g, b, y, m, r, M, d, A, d, n_out, u, n_out, alpha=1.05, drop_p=0.05, drop_seed=1, seed_dev=ctr))
    t_f0 = t(lambda: ops.layernorm_fwd_lora(x, g, b, y, m, r, M, d, A, d, n_out, u, n_out, alpha=1.05))
    print(f"n_out {n_out}: ln {t_ln:.1f} us  skinny {t_sk:.1f} us  fused {t_f:.1f} us  fused(no drop) {t_f0:.1f} us")
