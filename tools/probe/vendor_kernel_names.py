"""Which kernels the vendor libraries pick on the step's shapes (run under rocprofv3 --kernel-trace --stats; the Tensile kernel names carry the
macro tile, the MFMA shape and the workgroup size).  Every shape runs under an NVTX-free marker kernel: a torch.zeros(n) fill whose size encodes
the case index, so the trace can be cut per case."""
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
cases = [(96000, 1536, 512), (96000, 512, 512), (96000, 2048, 512), (96000, 512, 2048), (96000, 512, 1536), (96000, 6144, 512), (96000, 512, 6144),
         (384000, 512, 1536), (96000, 3840, 1280), (96000, 1280, 1280), (96000, 5120, 1280), (96000, 1280, 5120), (4096, 4096, 4096), (8192, 8192, 8192)]
for i, (M, N, K) in enumerate(cases):
    x = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * 0.04).half(); y = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3):
        torch.matmul(x, W.t(), out=y)
    torch.cuda.synchronize()
    print("case", i, M, N, K, flush=True)
for H in (8, 20):
    q, k, v = (torch.randn(64, H, 1500, 64, device=dev).half().requires_grad_(True) for _ in range(3))
    for _ in range(2):
        o = F.scaled_dot_product_attention(q, k, v, scale=1.0)
        o.backward(torch.ones_like(o))
    torch.cuda.synchronize()
