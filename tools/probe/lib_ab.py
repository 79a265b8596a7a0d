"""Same-process A/B of two BUILDS of the library (cdna guide rule 24: never rank builds by timings from different processes or boxes): the
tree's libneuspeech_hip.so against tools/probe/build/libns_prev.so (a build of an earlier commit, made with `git worktree`), interleaved
rounds on the same inputs.  CASE=attn_fwd|attn_bwd|gemm ...; outputs are also compared bit for bit."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from neuspeech1_amd import lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
new = L.load()
prev = C.CDLL(os.environ.get("PREV", os.path.join(ROOT, "tools", "probe", "build", "libns_prev.so")))
for name, (res, args) in L.SIGNATURES.items():
    if hasattr(prev, name):
        getattr(prev, name).restype, getattr(prev, name).argtypes = res, args


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(name, call, outs, rounds=6, n=10):
    """call(lib): one launch through `lib`; outs(): tensors to compare"""
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for tag, lib in (("prev", prev), ("new", new)):
        call(lib, st); call(lib, st)
        torch.cuda.synchronize()
        res[tag] = [t.clone() for t in outs()]
    same = all(torch.equal(a, b) for a, b in zip(res["prev"], res["new"]))
    best = {"prev": [], "new": []}
    for _ in range(rounds):
        for tag, lib in (("prev", prev), ("new", new)):
            best[tag].append(timed(lambda: call(lib, st), n))
    p, q = min(best["prev"]), min(best["new"])
    pm, qm = sorted(best["prev"])[rounds // 2], sorted(best["new"])[rounds // 2]
    print(f"{name:44s} prev {p:8.1f} us (median {pm:8.1f})   new {q:8.1f} us (median {qm:8.1f})   new/prev {q / p:.3f}   bitwise equal: {same}", flush=True)


def attn_case(B, H, S, what):
    d = H * 64
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = (torch.randn(B * S, 3 * d, device=dev, generator=g) * 0.5).half()
    O = torch.zeros(B * S, d, device=dev, dtype=torch.float16)
    LSE = torch.zeros(B, H, S, device=dev)
    common = dict(Q=qkv, K=(qkv, d), V=(qkv, 2 * d), O=O, B=B, H=H, Lq=S, Lk=S, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, causal=False, LSE=LSE)
    if what == "fwd":
        desc = ops._attn_desc(**common)
        ab(f"attn_fwd B={B} H={H} S={S}", lambda lib, st: L.check(lib.ns_attn_fwd(C.byref(desc), st)), lambda: (O, LSE))
        return
    ops.attn_fwd(**common)
    dO = (torch.randn(B * S, d, device=dev, generator=g) * 0.5).half()
    dqkv = torch.zeros(B * S, 3 * d, device=dev, dtype=torch.float16)
    Delta = torch.zeros(B, H, S, device=dev)
    ws = torch.zeros(ops.attn_bwd_workspace_bytes(B, H, S, S), device=dev, dtype=torch.uint8)
    desc = ops._attn_desc(**common, dO=dO, dQ=dqkv, dK=(dqkv, d), dV=(dqkv, 2 * d), Delta=Delta, lddo=d, lddq=3 * d, lddk=3 * d, lddv=3 * d, workspace=ws)
    ab(f"attn_bwd (one pass) B={B} H={H} S={S}", lambda lib, st: L.check(lib.ns_attn_bwd(C.byref(desc), st)), lambda: (dqkv,))


if __name__ == "__main__":
    for case in (sys.argv[1:] or ["attn_fwd"]):
        if case == "attn_fwd":
            attn_case(64, 8, 1500, "fwd")
            attn_case(64, 20, 1500, "fwd")
            attn_case(64, 8, 1472, "fwd")       # a multiple of 64: no clamped tile
            attn_case(3, 8, 700, "fwd")
        elif case == "attn_bwd":
            attn_case(64, 8, 1500, "bwd")
            attn_case(64, 20, 1500, "bwd")
