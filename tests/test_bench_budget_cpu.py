"""CPU: the accounting behind bench.py's `step_budget` / `roofline.algorithmic_bytes_per_launch` (round 6) -- the per-launch
algorithmic FLOP / byte formulas of the ops wrappers and the aggregation into classes, families, floors and totals.  No GPU: events are faked."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Ev:
    """stands in for a pair of torch.cuda.Events: elapsed_time() of the start event returns the milliseconds it was built with"""

    def __init__(self, ms=0.0):
        self.ms = ms

    def elapsed_time(self, other):
        return other.ms


def test_gemm_work_counts_every_operand_and_output_once():
    from neuspeech1_amd import ops
    M, N, K, r = 96000, 512, 2048, 32
    t = object()      # any non-None stands for a tensor here
    # fc2 + fp32 residual + adapter second product: A, W, u, sB, bias in; R32 in, H32 out
    fl, by = ops._gemm_work(dict(M=M, N=N, K=K, K2=r, A=t, am=ops.rowmap(K), B=t, bias=t, R32=t, H32=t))
    assert fl == 2.0 * M * N * (K + r)
    assert by == 2.0 * M * K + 2.0 * N * K + 2.0 * (M + N) * r + 4.0 * M * N * 2 + 4.0 * N
    # fc1 + GELU + saved gelu' + side product (8 column tiles of 256)
    fl, by = ops._gemm_work(dict(M=M, N=2048, K=512, A=t, am=ops.rowmap(512), B=t, C16=t, G16=t, side_B=t, side_n=32))
    assert fl == 2.0 * M * 2048 * (512 + 32)
    assert by == 2.0 * M * 512 + 2.0 * 2048 * 512 + 2.0 * M * 2048 * 2 + 2.0 * 32 * 2048 + 4.0 * 8 * M * 32
    # conv-as-GEMM over a halo image: overlapping k = 3 windows count the image once (segments x seg_stride), FLOPs on the algorithmic length
    B_, T, C = 64, 6000, 208
    am = ops.rowmap(C, T, (T + 2) * C)
    fl, by = ops._gemm_work(dict(M=B_ * T, N=512, K=3 * C, k_alg=3 * 208, A=t, am=am, B=t, C16=t))
    assert fl == 2.0 * B_ * T * 512 * 3 * 208
    assert by == 2.0 * B_ * (T + 2) * C + 2.0 * 512 * 3 * C + 2.0 * B_ * T * 512
    # weight gradient (TN): both reduction-major operands once, the fp32 result once
    fl, by = ops._gemm_work(dict(M=32, N=512, K=96000, flags=ops.NS_GEMM_TN, A=t, B=t, C32=t))
    assert fl == 2.0 * 32 * 512 * 96000 and by == 2.0 * 96000 * (32 + 512) + 4.0 * 32 * 512
    # dispatch families mirror csrc/ns_gemm.hip
    assert ops._gemm_kind(dict(M=96000, N=512)) == "nt256" and ops._gemm_kind(dict(M=2816, N=512)) == "nt128"
    assert ops._gemm_kind(dict(M=96000, N=32)) == "nt32" and ops._gemm_kind(dict(M=640, N=2048)) == "nt32"
    assert ops._gemm_kind(dict(M=32, N=512, flags=ops.NS_GEMM_TN)) == "tn"
    assert ops._gemm_epi(dict(H32=t, K2=32)) == "res+k2" and ops._gemm_epi(dict(flags=ops.NS_GEMM_GELU, side_B=t)) == "gelu+side"
    assert ops._gemm_epi(dict(flags=ops.NS_GEMM_DGELU, K2=32, drop_p=0.05)) == "dgelu+k2+drop"


def test_attention_work_is_two_products_forward_and_five_backward():
    from neuspeech1_amd import ops
    kw = dict(B=64, H=8, Lq=1500, Lk=1500, causal=False)
    ff, fb = ops._attn_work(kw, False)
    bf, bb = ops._attn_work(kw, True)
    assert ff == 4.0 * 64 * 8 * 1500 * 1500 * 64 and bf == 2.5 * ff
    assert fb == 2.0 * 64 * 8 * 64 * (2 * 1500 + 2 * 1500) + 4.0 * 64 * 8 * 1500 and bb == 2.0 * fb + 4.0 * 64 * 8 * 1500
    assert ops._attn_work(dict(kw, causal=True), False)[0] == 0.5 * ff


def test_step_budget_aggregates_classes_families_and_floors(monkeypatch):
    import bench
    monkeypatch.setattr(bench, "_counter_tables", lambda: ({"void ns_gemm_p8s_kernel<false, 0, true>": (12, 0.9e9), "ln_bwd_kernelILi8": (60, 0.33e9)},
                                                           {"void ns_gemm_p8s_kernel<false, 0, true>": (12.0, 2.0e7, 0.29)}, 12))
    recs = []
    # six dominant GEMM launches of 0.2 ms, 150 GFLOP and 0.4 GB each (MFMA-bound at the sustained rate), 30 LayerNorm backward launches of 0.05 ms and 0.3 GB (HBM-bound)
    for _ in range(6):
        recs.append(("gemm nt256 plain K=512 N=1536", 150e9, 0.4e9, _Ev(), _Ev(0.2)))
    for _ in range(30):
        recs.append(("layernorm_bwd", 0.0, 0.3e9, _Ev(), _Ev(0.05)))
    recs.append(("add_i32", 0.0, 0.0, _Ev(), _Ev(0.006)))
    b = bench.step_budget(recs, step_ms=2.7, sustained=1868.4)
    c = b["classes"]["gemm nt256 plain K=512 N=1536"]
    assert c["launches"] == 6 and abs(c["ms"] - 1.2) < 1e-9 and c["bound"] == "mfma"
    assert abs(c["tflops"] - 6 * 150e9 / 1.2e-3 / 1e12) < 0.1 and abs(c["floor_ms_sustained"] - 6 * 150e9 / 1868.4e12 * 1e3) < 1e-3
    assert abs(c["floor_ms_spec"] - max(6 * 0.4e9 / 8e12, 6 * 150e9 / 2.5e15) * 1e3) < 1e-3
    ln = b["classes"]["layernorm_bwd"]
    assert ln["bound"] == "hbm" and abs(ln["floor_ms_sustained"] - 30 * 0.3e9 / 6.3e12 * 1e3) < 1e-3 and abs(ln["tb_per_s"] - 30 * 0.3e9 / 1.5e-3 / 1e12) < 0.01
    fam = b["families"]["gemm 256x256 (dominant)"]
    # the counter file holds 12 dominant launches = 2 steps of 6: 12 x 0.9 GB / 2 steps
    assert fam["launches"] == 6 and abs(fam["counter_gb"] - 5.4) < 1e-6 and abs(fam["counter_over_algorithmic"] - 5.4 / 2.4) < 0.01 and fam["mfma_busy"] == 0.29
    assert b["families"]["LayerNorm backward"]["counter_gb"] == round(60 * 0.33e9 / 2 / 1e9, 3)
    assert "clears / packing / rest" in b["families"]
    tot = b["total"]
    assert tot["launches"] == 37 and abs(tot["ms"] - (1.2 + 1.5 + 0.006)) < 1e-6
    assert abs(tot["floor_ms_sustained"] - (c["floor_ms_sustained"] + ln["floor_ms_sustained"])) < 2e-3      # a step is a chain: floors ADD
    assert b["rates"]["mfma_sustained_tflops"] == 1868.4 and b["step_ms_replayed"] == 2.7


def test_sustained_peak_comes_from_the_committed_yardstick():
    import bench
    s = bench.sustained_mfma_tflops()
    assert s is not None and 1500.0 < s < 2300.0       # back-to-back 16x16x32 fp16 MFMAs on random operands: ~0.75 of the 2.5 PFLOP/s spec peak


def test_encoder_ceiling_follows_from_the_committed_yardstick():
    """profiles/r6_encoder_ceiling.json is what tools/probe/encoder_ceiling.py derives from profiles/r6_yardstick.json:
    with the faster of {vendor, this library} on every MFMA launch of the encoder's forward + backward and everything else free, the floor is above
    what north_star's 40 % MFMA target allows."""
    import json
    import runpy
    committed = json.load(open(os.path.join(ROOT, "profiles", "r6_encoder_ceiling.json")))
    before = open(os.path.join(ROOT, "profiles", "r6_encoder_ceiling.json")).read()
    try:
        runpy.run_path(os.path.join(ROOT, "tools", "probe", "encoder_ceiling.py"), run_name="__main__")
        again = json.load(open(os.path.join(ROOT, "profiles", "r6_encoder_ceiling.json")))
    finally:
        open(os.path.join(ROOT, "profiles", "r6_encoder_ceiling.json"), "w").write(before)
    assert again == committed
    assert committed["floor_ms"] > committed["ms_allowed_by_40_percent"] > 13.0
    assert 0.25 < committed["mfma_frac_at_floor"] < 0.40
    kinds = {r["launch"]: r["kernel"] for r in committed["per_layer_us"]}
    assert kinds["attention forward"] == "ns" and kinds["attention backward"] == "ns"      # ours is the faster attention in both directions
