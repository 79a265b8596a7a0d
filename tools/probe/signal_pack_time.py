"""ns_signal_pack at the bench shape (64, 208, 6000) and the decode shape (128, 273, 6000): microseconds and TB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neuspeech1_amd import ops
dev = torch.device("cuda:0")
for B, ch in ((64, 208), (128, 273)):
    T, Cp = 6000, (ch + 63) // 64 * 64
    x = torch.randn(B, ch, T, device=dev)
    out = torch.zeros(B, T + 2, Cp, device=dev, dtype=torch.float16)
    for _ in range(3): ops.signal_pack(x, out, B, ch, T, Cp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.signal_pack(x, out, B, ch, T, Cp)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    by = x.numel() * 4 + B * T * Cp * 2
    print(f"signal_pack B {B} ch {ch}: {us:.1f} us  {by / us / 1e6:.2f} TB/s")
