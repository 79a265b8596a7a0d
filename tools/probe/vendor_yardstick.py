"""External yardstick: on ONE box, in ONE process, interleaved rounds (cdna guide rule 24), random data
(rule 25):
  (a) the vendor GEMM (torch.matmul fp16 -> hipBLASLt / rocBLAS) against ns_gemm (plain fp16 epilogue) on the step's dominant shapes and
      on whisper-large-v2's;
  (b) the vendor attention (F.scaled_dot_product_attention, every backend that runs) forward and backward against ns_attn_fwd / ns_attn_bwd
      at (64, 8, 1500, 64) and (64, 20, 1500, 64);
  (c) tools/probe/build/mfma_clock: the clock and MFMA rate the chip holds under a sustained register-only / LDS-fed fp16 MFMA load.
Writes gpurun_out/r6_yardstick.json (copied to profiles/ by hand)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from neuspeech1_amd import ops  # noqa: E402
from neuspeech1_amd.ops import rowmap  # noqa: E402

dev = torch.device("cuda:0")
F16, F32 = torch.float16, torch.float32
PEAK = 2.5e15
out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "gemm": [], "attention": [], "mfma_clock": []}
QUICK = bool(int(os.environ.get("QUICK", "0")))


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ab(fns: dict, rounds=5, n=8):
    """min and median ms per variant over interleaved rounds"""
    res = {k: [] for k in fns}
    for k, fn in fns.items():          # warm-up (also hipBLASLt's heuristic / torch's workspace allocation)
        try:
            fn(); fn()
            torch.cuda.synchronize()
        except Exception as e:          # a backend that refuses the shape
            res[k] = None
            print(f"   [{k}] refused: {type(e).__name__}: {str(e)[:120]}", flush=True)
    for _ in range(rounds):
        for k, fn in fns.items():
            if res[k] is not None:
                res[k].append(timed(fn, n))
    return {k: (None if v is None else {"min_ms": min(v), "median_ms": sorted(v)[len(v) // 2]}) for k, v in res.items()}


# ---------------------------------------------------------------- (c) first: the sustained clock, on a cool chip and again at the end
def mfma_clock(tag):
    exe = os.path.join(ROOT, "tools", "probe", "build", "mfma_clock")
    if not os.path.exists(exe):
        src = os.path.join(ROOT, "tools", "probe", "mfma_clock.hip")
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, src], check=True)
    r = subprocess.run([exe, "1.0" if QUICK else "2.5"], capture_output=True, text=True, timeout=600)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            d["when"] = tag
            d["frac_of_spec_peak"] = d["tflops"] * 1e12 / PEAK
            out["mfma_clock"].append(d)
            print(f"  {d['probe']:40s} {d['waves_per_simd']} w/SIMD  {d['tflops']:7.1f} TFLOP/s  clock {d['clock_ghz_median']:.3f} GHz "
                  f"({d['clock_ghz_min']:.3f}-{d['clock_ghz_max']:.3f})  {d['cycles_per_mfma_per_simd']:.2f} cyc/MFMA/SIMD", flush=True)
    if r.returncode != 0:
        print("  mfma_clock failed:", r.stderr[-400:], flush=True)


# ---------------------------------------------------------------- (a) GEMM
def gemm_cases():
    M = 96000
    base = [  # (name, M, N, K): the step's launches
        ("base q|k|v", M, 1536, 512), ("base out_proj / dgrads K=512", M, 512, 512), ("base fc1 / fc2 dgrad", M, 2048, 512),
        ("base fc2 / fc1 dgrad", M, 512, 2048), ("base q|k|v dgrad / conv2", M, 512, 1536), ("base cross K|V stacked", M, 6144, 512),
        ("base cross K|V dgrad", M, 512, 6144), ("base conv1.2", 384000, 512, 1536), ("base conv1.0 (208 ch)", 384000, 512, 624),
        ("square-ish 1536", M, 1536, 1536), ("square-ish 2048", M, 2048, 2048), ("square-ish 6144 x 2048", M, 6144, 2048),
    ]
    lv2 = [("lv2 q|k|v", M, 3840, 1280), ("lv2 out_proj", M, 1280, 1280), ("lv2 fc1", M, 5120, 1280), ("lv2 fc2", M, 1280, 5120),
           ("lv2 q|k|v dgrad", M, 1280, 3840)]
    guide = [("guide 4096^3", 4096, 4096, 4096), ("guide 8192^3", 8192, 8192, 8192)]
    return (base[:4] + lv2[:2] + guide[:1]) if QUICK else base + lv2 + guide


def run_gemm():
    g = torch.Generator(device=dev).manual_seed(3)
    for name, M, N, K in gemm_cases():
        x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(F16)
        W = (torch.randn(N, K, device=dev, generator=g) * 0.04).to(F16)
        Wt = W.t().contiguous()
        y = torch.empty(M, N, device=dev, dtype=F16)
        y2 = torch.empty(M, N, device=dev, dtype=F16)
        fns = {
            "vendor_nt": lambda: torch.matmul(x, W.t(), out=y),        # A (M, K) x W (N, K)^T: ns_gemm's operand layout
            "vendor_nn": lambda: torch.matmul(x, Wt, out=y),           # W stored (K, N)
            "ns_gemm": lambda: ops.gemm(A=x, am=rowmap(K), K=K, B=W, ldb=K, M=M, N=N, C16=y2, c16m=rowmap(N)),
        }
        r = ab(fns)
        # same numbers? (fp32 accumulate on both sides; order differs)
        torch.matmul(x, W.t(), out=y); fns["ns_gemm"](); torch.cuda.synchronize()
        err = (y.float() - y2.float()).abs().max().item()
        flop = 2.0 * M * N * K
        row = {"name": name, "M": M, "N": N, "K": K, "gflop": flop / 1e9, "max_abs_diff_vs_vendor": err}
        for k, v in r.items():
            row[k] = None if v is None else {**v, "tflops": flop / (v["min_ms"] * 1e-3) / 1e12, "frac_of_spec_peak": flop / (v["min_ms"] * 1e-3) / PEAK}
        vend = max((row[k]["tflops"] for k in ("vendor_nt", "vendor_nn") if row[k]), default=None)
        row["ns_over_best_vendor"] = None if vend is None else row["ns_gemm"]["tflops"] / vend
        out["gemm"].append(row)
        f = lambda k: "   -   " if row[k] is None else f"{row[k]['min_ms'] * 1e3:7.1f} us {row[k]['tflops']:6.0f} TF"
        print(f"  {name:32s} M {M:6d} N {N:5d} K {K:5d} | vendor NT {f('vendor_nt')} | vendor NN {f('vendor_nn')} | ns_gemm {f('ns_gemm')} | "
              f"ns / vendor {row['ns_over_best_vendor']:.2f}  (max |diff| {err:.3g})", flush=True)
        del x, W, Wt, y, y2


# ---------------------------------------------------------------- (b) attention
def run_attention():
    from torch.nn.attention import SDPBackend, sdpa_kernel
    S = 1500
    for name, B, H in (("base (64, 8, 1500, 64)", 64, 8), ("large-v2 (64, 20, 1500, 64)", 64, 20)):
        if QUICK and H == 20:
            continue
        d = H * 64
        g = torch.Generator(device=dev).manual_seed(1)
        qkv = (torch.randn(B * S, 3 * d, device=dev, generator=g) * 0.5).to(F16)
        dO = (torch.randn(B * S, d, device=dev, generator=g) * 0.5).to(F16)
        # ---- ours: heads read in place from the fused (rows, 3d) buffer
        O = torch.zeros(B * S, d, device=dev, dtype=F16)
        LSE = torch.zeros(B, H, S, device=dev)
        dqkv = torch.zeros(B * S, 3 * d, device=dev, dtype=F16)
        Delta = torch.zeros(B, H, S, device=dev)
        ws = torch.zeros(ops.attn_bwd_workspace_bytes(B, H, S, S), device=dev, dtype=torch.uint8)
        common = dict(Q=qkv, K=(qkv, d), V=(qkv, 2 * d), O=O, B=B, H=H, Lq=S, Lk=S, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, causal=False, LSE=LSE)
        bw = dict(dO=dO, dQ=dqkv, dK=(dqkv, d), dV=(dqkv, 2 * d), Delta=Delta, lddo=d, lddq=3 * d, lddk=3 * d, lddv=3 * d)
        ops.attn_fwd(**common)
        # ---- vendor: (B, H, S, 64) views of the same buffer (strided), and contiguous copies (what a stock model hands over after its transposes)
        v4 = qkv.view(B, S, 3, H, 64)
        qs, ks, vs = (v4[:, :, i].transpose(1, 2) for i in range(3))
        qc, kc, vc = (t.contiguous() for t in (qs, ks, vs))
        dOc = dO.view(B, S, H, 64).transpose(1, 2).contiguous()
        fwd, bwd = {}, {}
        fwd["ns_attn_fwd"] = lambda: ops.attn_fwd(**common)
        bwd["ns_attn_bwd (one pass)"] = lambda: ops.attn_bwd(**common, **bw, workspace=ws)
        bwd["ns_attn_bwd (two passes)"] = lambda: ops.attn_bwd(**common, **bw)
        backends = {"flash": SDPBackend.FLASH_ATTENTION, "efficient": SDPBackend.EFFICIENT_ATTENTION, "math": SDPBackend.MATH}
        if QUICK:
            backends.pop("math")
        for bn, be in backends.items():
            for lay, (q_, k_, v_) in (("contig", (qc, kc, vc)), ("strided", (qs, ks, vs))):
                if bn == "math" and lay == "strided":
                    continue

                def f(q_=q_, k_=k_, v_=v_, be=be):
                    with sdpa_kernel(be), torch.no_grad():
                        return F.scaled_dot_product_attention(q_, k_, v_, scale=1.0)   # ns_attn takes q pre-scaled (folded into the weights)

                def fb(q_=q_, k_=k_, v_=v_, be=be):
                    q1, k1, v1 = (t.detach().requires_grad_(True) for t in (q_, k_, v_))
                    with sdpa_kernel(be):
                        o = F.scaled_dot_product_attention(q1, k1, v1, scale=1.0)
                    o.backward(dOc)

                fwd[f"sdpa {bn} {lay}"] = f
                bwd[f"sdpa {bn} {lay} fwd+bwd"] = fb
        rf, rb = ab(fwd, rounds=4, n=4), ab(bwd, rounds=4, n=3)
        fl = 4.0 * B * H * S * S * 64
        row = {"name": name, "B": B, "H": H, "S": S, "fwd_gflop": fl / 1e9, "bwd_gflop_algorithmic": 2.5 * fl / 1e9, "fwd": {}, "bwd": {}}
        for k, v in rf.items():
            row["fwd"][k] = None if v is None else {**v, "tflops": fl / (v["min_ms"] * 1e-3) / 1e12}
        for k, v in rb.items():
            if v is None:
                row["bwd"][k] = None
                continue
            ms = v["min_ms"]
            if k.startswith("sdpa"):        # backward alone = (fwd + bwd) - the same backend's forward
                fk = k.replace(" fwd+bwd", "")
                if rf.get(fk):
                    ms = ms - rf[fk]["min_ms"]
            row["bwd"][k] = {**v, "bwd_only_ms": ms, "tflops_algorithmic": 2.5 * fl / (ms * 1e-3) / 1e12}
        out["attention"].append(row)
        print(f"  {name}", flush=True)
        for k, v in row["fwd"].items():
            print(f"    fwd  {k:34s} " + ("refused" if v is None else f"{v['min_ms']:8.3f} ms  {v['tflops']:6.0f} TF ({v['tflops'] / 2500:.3f} of peak)"), flush=True)
        for k, v in row["bwd"].items():
            print(f"    bwd  {k:34s} " + ("refused" if v is None else f"{v['bwd_only_ms']:8.3f} ms  {v['tflops_algorithmic']:6.0f} TF algorithmic"), flush=True)
        # agreement of the two backward passes (ours against the vendor's best backend), loosely: both are fp16
        try:
            q1, k1, v1 = (t.detach().requires_grad_(True) for t in (qc, kc, vc))
            o = F.scaled_dot_product_attention(q1, k1, v1, scale=1.0)
            o.backward(dOc)
            ops.attn_bwd(**common, **bw, workspace=ws)
            torch.cuda.synchronize()
            mine = dqkv.view(B, S, 3, H, 64)
            row["max_abs_diff_dq_dk_dv"] = [(mine[:, :, i].transpose(1, 2).float() - t.grad.float()).abs().max().item() for i, t in enumerate((q1, k1, v1))]
            print("    max |ours - vendor| dq, dk, dv:", ["%.3g" % e for e in row["max_abs_diff_dq_dk_dv"]], flush=True)
        except Exception as e:
            print("    agreement check skipped:", type(e).__name__, str(e)[:100], flush=True)
        del qkv, dO, O, dqkv, qc, kc, vc, dOc, ws


if __name__ == "__main__":
    what = sys.argv[1:] or ["clock", "gemm", "attention", "clock2"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for w in what:
        print(f"== {w}", flush=True)
        if w == "clock":
            mfma_clock("start")
        elif w == "clock2":
            mfma_clock("end")
        elif w == "gemm":
            run_gemm()
        elif w == "attention":
            run_attention()
        with open(os.path.join(ROOT, "gpurun_out", "r6_yardstick.json"), "w") as fh:
            json.dump(out, fh, indent=1)
