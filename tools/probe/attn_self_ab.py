"""ns_attn_decode on the ancestry layout (the decode loop's self-attention): one wave per (row, head) against the four-wave kernel,
at the eval leg's shapes (128 rows greedy, 640 rows beam-5; 8 heads; 6 layers' caches rotated so that rows come from HBM / MALL
as in the loop).  Prints us per launch for both forms and the largest difference of the outputs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuspeech1_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
H, d, Lmax, NL, REP = 8, 64, 68, 6, 20
for rows in (128, 640):
    g = torch.Generator(device=dev).manual_seed(rows)
    Q = [(torch.randn(rows, 3 * H * d, device=dev, generator=g) * 0.6).half() for _ in range(NL)]
    cache = [(torch.randn(Lmax * rows, 2 * H * d, device=dev, generator=g) * 0.7).half() for _ in range(NL)]
    grp = torch.arange(rows, device=dev) // max(1, rows // 128)
    anc = (grp.unsqueeze(1) * max(1, rows // 128) + torch.randint(0, max(1, rows // 128), (rows, Lmax), device=dev, generator=g)).int()
    anc[:, Lmax - 1] = torch.arange(rows, device=dev, dtype=torch.int32)
    O = torch.empty(rows, H * d, device=dev, dtype=torch.float16)
    for Lk in (8, 36, 68):
        klen = torch.tensor([Lk], device=dev, dtype=torch.int32)
        res, outs = {}, {}
        for form in (2, 0):
            so.ns_debug_set_ad_self(form)

            def fn(i):
                ops.attn_decode(Q=Q[i], K=cache[i], V=(cache[i], H * d), O=O, groups=rows, nq=1, H=H, Lk=Lmax, Lk_max=Lmax, ldq=3 * H * d,
                                ldk=2 * H * d, ldv=2 * H * d, ldo=H * d, anc=anc, anc_ld=Lmax, kv_pos_stride=rows, kv_len_dev=klen,
                                Knew=(Q[i], H * d), Vnew=(Q[i], 2 * H * d), ldnew=3 * H * d)
            for i in range(NL):
                fn(i)
            outs[form] = O.float().clone()
            lst = ops.LaunchList()                 # replayed at ~1 us per launch: the clock sees the kernels, not ctypes
            with ops.recording(lst):
                for i in range(NL):
                    fn(i)
            lst.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP):
                lst.replay()
            e1.record()
            torch.cuda.synchronize()
            res[form] = e0.elapsed_time(e1) * 1e3 / (REP * NL)
        so.ns_debug_set_ad_self(1)
        print(f"rows {rows:4d} Lk {Lk:3d}: wave per head {res[2]:6.2f} us   four waves {res[0]:6.2f} us   ratio {res[2] / res[0]:.2f}   "
              f"max |diff| {(outs[2] - outs[0]).abs().max().item():.2e}", flush=True)
