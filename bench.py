#!/usr/bin/env python3
"""Headline benchmark: train samples/sec of the Whisper-MEG hot path (whisper-base, 208-ch, bs64/GPU,
fp16 LoRA r=32 training step = forward + backward + RCCL gradient all-reduce + clip + AdamW).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  Inputs are synthetic (B,208,6000) MEG
tensors already resident in HBM; weights are seeded random-init of the whisper-base architecture.

Beside the contract's keys the line carries: `roofline` (dominant GEMM class: achieved TFLOP/s by HIP events, `frac` of the 2.5 PFLOP/s spec peak and
`frac_of_sustained` of the rate the chip holds on random operands, counter `traffic` against `algorithmic_bytes_per_launch`), `encoder_fwd_bwd`
(the quantity north_star's 40 % target is stated on), `step_budget` (round 6: per kernel class and family -- launches, ms, algorithmic FLOP / bytes,
achieved rates, floors at the spec and at the sustained rates, counter bytes, mfma_busy; their sum of floors against the step), `cpu_baseline`
(+ `cpu_baseline_port`, `torch_rocm_reference_object`), `eval` (tokens/s, decode-step HBM roofline, first / steady call), `large_v2`, `dp` at N > 1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC: RCCL across processes needs this (already exported on the GPU box)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# algorithmic work per training sample, fwd+bwd: whisper-base, L=32, LoRA r=32
GFLOP_PER_SAMPLE = {208: 239.67, 273: 242.06}
# encoder-only forward + backward (conv stem + 6 layers + adapters; SURVEY §8d: encoder forward 100.49 GFLOP, backward =
# the whole backward 126.03 minus the decoder's dgrad 13.15): what north_star's ">= 40 % MFMA on whisper-base encoder
# fwd+bwd" is measured on
ENC_GFLOP_PER_SAMPLE = {208: 213.37, 273: 215.76}
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16, MI355X_MICROARCH.md
# the dominant kernel: the phase-interleaved 256 x 256 GEMM in its two forms -- one tile per workgroup (ns_gemm_p8_kernel: launches of < 700
# tiles, the position-row epilogue) and persistent (ns_gemm_p8s_kernel: >= 700 tiles, NS_P8S_MIN_TILES in csrc/ns_gemm.hip, part of the
# counter file's source hash) -- one tile arithmetic, one "nt256" class in the event-timed leg; in the rocprof stats its average launch = all
# ns_gemm_p8*_kernel rows together.  Since round 5 the class has 52 launches per step (68 before): out_proj + LayerNorm run as ns_gemm_ln
# ("rowln" in step_share), the six decoder layers' cross K|V projections and their input gradients as one launch each
DOMINANT = "ns_gemm_p8_kernel+ns_gemm_p8s_kernel"
PMC_FILE = "profiles/r6_pmc_traffic.json"   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile.sh pmc), hash-guarded
DECODE_PMC_FILE = "profiles/r6_decode_pmc_traffic.json"   # the same two passes over tools/bench_decode.py (tools/profile.sh decode_pmc)
LV2_GFLOP_PER_SAMPLE = 5630.0   # whisper-large-v2, 273-ch, fwd+bwd
# round 6: the external yardstick (tools/probe/vendor_yardstick.py -> profiles/r6_yardstick.json): the MFMA rate and HBM rate this chip HOLDS.
# `frac` stays against the spec peaks (2.5 PFLOP/s, 8 TB/s); `frac_of_sustained` is reported BESIDE it, never instead of it.
YARDSTICK_FILE = "profiles/r6_yardstick.json"
HBM_PEAK_GBS, HBM_ACHIEVABLE_GBS = 8000.0, 6300.0      # spec; float4 copy (MI355X_MICROARCH.md)
SQ_FILE = "profiles/r6_pmc_SQ_by_kernel.tsv"


def sustained_mfma_tflops():
    """the fp16 MFMA rate the chip sustains on RANDOM operands held in registers, every CU busy for seconds (back-to-back
    v_mfma_f32_16x16x32_f16, two waves per SIMD: the dominant GEMM's shape and occupancy), from the committed yardstick run; None without it"""
    try:
        d = json.load(open(os.path.join(ROOT, YARDSTICK_FILE)))
        c = [r["tflops"] for r in d["mfma_clock"] if r["probe"] == "mfma_16x16x32_f16 regs random" and r["waves_per_simd"] == 2]
        return round(sum(c) / len(c), 1) if c else None
    except Exception:
        return None


# step_budget families: (family, class prefixes of ops.STEP_PROFILE, kernel-name fragments of the rocprof files)
FAMILIES = [
    ("gemm 256x256 (dominant)", ("gemm nt256",), ("ns_gemm_p8s_kernel", "ns_gemm_p8_kernel")),
    ("gemm + LayerNorm rows", ("gemm rowln",), ("gemm_ln_kernel",)),
    ("gemm 128x128 / 64x128 ring", ("gemm nt128",), ("ns_gemm_ring_kernel", "ns_gemm_ring256_kernel")),
    ("gemm skinny / small", ("gemm nt32",), ("ns_gemm_skinny_kernel", "ns_gemm_kernel", "ns_gemm_smallm")),
    ("weight gradients (TN)", ("gemm tn",), ("ns_gemm_tn_kernel", "ns_gemm_tn256_kernel")),
    ("attention backward, one pass", ("attn_bwd one pass",), ("attn_bwd1_kernel",)),
    ("attention forward", ("attn_fwd",), ("attn_fwd_kernel", "attn_fewq")),
    ("attention backward, decoder", ("attn_bwd",), ("attn_bwd_fewq_kernel", "attn_fewq_dq_reduce_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel")),
    ("LayerNorm backward", ("layernorm_bwd",), ("ln_bwd_kernel",)),
    ("LayerNorm forward", ("layernorm_fwd",), ("ln_fwd_kernel",)),
    ("adapter up-projection backward", ("lora_bwd_dudb",), ("lora_bwd_dudb_kernel", "lora_bwd_reduce_kernel")),
    ("side-product reduce", ("side_reduce",), ("side_reduce_kernel",)),
    ("loss", ("cross_entropy",), ("ce_kernel",)),
    ("optimizer", ("adamw_step", "grad_norm", "cast_jobs"), ("adamw_kernel", "norm_partial_kernel", "norm_final", "cast_jobs_kernel")),
    ("clears / packing / rest", ("",), ()),
]


def _counter_tables():
    """per kernel: HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, PMC_FILE) and mfma_busy (SQ_FILE), both hash-guarded; {} when stale"""
    traffic, busy, steps_in_file = {}, {}, None
    try:
        from tools.kernel_hash import DOMINANT_SOURCES, source_hash
        d = json.load(open(os.path.join(ROOT, PMC_FILE)))
        if d.get("kernel_source_sha256_16") == source_hash(DOMINANT_SOURCES):
            traffic = {k: (v["launches"], v["hbm_bytes_per_launch"]) for k, v in d["kernels"].items()}
            steps_in_file = d.get("launches")
    except Exception:
        pass
    try:
        import csv
        with open(os.path.join(ROOT, SQ_FILE)) as f:
            for r in csv.DictReader(f, delimiter="\t"):
                busy[r["kernel"]] = (float(r["launches"]), float(r["SQ_BUSY_CYCLES_per_launch"]), float(r["mfma_busy"]))
    except Exception:
        pass
    return traffic, busy, steps_in_file


def step_budget(recs, step_ms, sustained):
    """one eager step, every launch bracketed by HIP events (ops.STEP_PROFILE).  Per class and per family: launches, time,
    algorithmic FLOP and bytes, achieved rates, the floor max(bytes / HBM rate, FLOP / MFMA rate) at the spec peaks and at the sustained rates,
    which of the two bounds it; per family also the fabric-side counter bytes and mfma_busy of the committed rocprofv3 passes."""
    cls = {}
    for c, fl, by, e0, e1 in recs:
        a = cls.setdefault(c, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += fl
        a[3] += by
    sus_f = (sustained or MFMA_PEAK_TFLOPS) * 1e12

    def row(n, ms, fl, by):
        f_spec = max(by / (HBM_PEAK_GBS * 1e9), fl / (MFMA_PEAK_TFLOPS * 1e12)) * 1e3
        t_h, t_m = by / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3, fl / sus_f * 1e3
        return {"launches": n, "ms": round(ms, 3), "gflop": round(fl / 1e9, 1), "algorithmic_gb": round(by / 1e9, 3),
                "tflops": round(fl / max(ms, 1e-9) / 1e9, 1), "tb_per_s": round(by / max(ms, 1e-9) / 1e9, 2),
                "floor_ms_spec": round(f_spec, 3), "floor_ms_sustained": round(max(t_h, t_m), 3), "bound": "hbm" if t_h >= t_m else "mfma"}
    classes = {c: row(*a) for c, a in sorted(cls.items(), key=lambda kv: -kv[1][1])}
    traffic, busy, dom_launches = _counter_tables()
    fams, used = {}, set()
    nt256_launches = sum(a[0] for c, a in cls.items() if c.startswith("gemm nt256"))
    file_steps = (dom_launches / nt256_launches) if (dom_launches and nt256_launches) else None
    for name, prefixes, frags in FAMILIES:
        mine = [c for c in cls if c not in used and any(c.startswith(p) for p in prefixes)]
        used.update(mine)
        if not mine:
            continue
        n, ms, fl, by = (sum(cls[c][i] for c in mine) for i in range(4))
        r = row(n, ms, fl, by)
        if frags and traffic and file_steps:
            tb = sum(v[0] * v[1] for k, v in traffic.items() if any(f in k for f in frags))
            r["counter_gb"] = round(tb / file_steps / 1e9, 3)
            r["counter_over_algorithmic"] = round(tb / file_steps / max(by, 1.0), 2)
        if frags and busy:
            w = [(v[0] * v[1], v[2]) for k, v in busy.items() if any(f in k for f in frags)]
            if w and sum(x for x, _ in w) > 0:
                r["mfma_busy"] = round(sum(x * b for x, b in w) / sum(x for x, _ in w), 3)
        fams[name] = r
    tot_ms = sum(a[1] for a in cls.values())
    tot = row(sum(a[0] for a in cls.values()), tot_ms, sum(a[2] for a in cls.values()), sum(a[3] for a in cls.values()))
    tot["floor_ms_spec"] = round(sum(v["floor_ms_spec"] for v in classes.values()), 3)             # SUM of the classes' floors: a step is a chain
    tot["floor_ms_sustained"] = round(sum(v["floor_ms_sustained"] for v in classes.values()), 3)
    tot.pop("bound")
    if traffic and file_steps:
        tot["counter_gb"] = round(sum(v[0] * v[1] for v in traffic.values()) / file_steps / 1e9, 2)
    return {"source": "one eager step, HIP events around every entry-point launch (a replayed step has no gaps between them: step time = "
                      "kernel time); algorithmic bytes = every operand and output once; counters from " + PMC_FILE + " / " + SQ_FILE,
            "rates": {"hbm_spec_gb_s": HBM_PEAK_GBS, "hbm_achievable_gb_s": HBM_ACHIEVABLE_GBS, "mfma_spec_tflops": MFMA_PEAK_TFLOPS,
                      "mfma_sustained_tflops": sustained},
            "step_ms_replayed": round(step_ms, 3), "total": tot, "families": fams, "classes": classes}


def cpu_baseline(dims, r, alpha):
    """Oracle (CPU port of the reference path: fp32 torch, all host cores) on a bounded sample: B=2, 1 warm-up +
    2 timed forward+backward passes with the same trainable set (LoRA + conv stem)."""
    import torch
    from neuspeech1_amd.weights import make_lora_state, make_state_dict, synth_batch
    from oracle import whisper_meg_oracle as O
    # fp32 torch on the host: 32 threads is where this workload stops scaling (256 oversubscribed threads ran
    # 50x slower on the first GPU box); `cores` reports the threads actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = make_state_dict(dims, 42)
    lora = make_lora_state(dims, r)
    B = 2
    x, labels = synth_batch(dims, B, 1234)
    O.loss_and_grads(sd, lora, x, labels, dims, alpha / r)
    t0 = time.perf_counter()
    n = 2
    for _ in range(n):
        O.loss_and_grads(sd, lora, x, labels, dims, alpha / r)
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 4), "unit": "samples/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle fp32 fwd+bwd, whisper-base {dims.ch}-ch, B={B}, {n} timed passes after 1 warm-up"}


def _pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    need separate profiler runs, so they cannot be collected inside this process): PMC_FILE, written by
    tools/profile.sh pmc + tools/pmc_summary.py.  The file records the hash of the kernel's comment- and whitespace-stripped
    source (tools/kernel_hash.py); a file measured on a different kernel revision is refused (traffic: null) rather than
    reported, and tests/test_profiles_cpu.py fails on such a file before it can reach the driver."""
    try:
        from tools.kernel_hash import DOMINANT_SOURCES, source_hash
        d = json.load(open(os.path.join(ROOT, PMC_FILE)))
        if not kernel or d.get("kernel") != kernel or d.get("kernel_source_sha256_16") != source_hash(DOMINANT_SOURCES):
            return None
        return round(d["hbm_bytes_per_launch"])
    except Exception:
        return None


def _decode_pmc_traffic():
    """HBM bytes per decode STEP (greedy / beam-5) from the committed FETCH_SIZE / WRITE_SIZE passes over
    tools/bench_decode.py (tools/profile.sh decode_pmc + tools/pmc_decode_summary.py), hash-guarded like _pmc_traffic"""
    try:
        from tools.kernel_hash import DECODE_SOURCES, source_hash
        d = json.load(open(os.path.join(ROOT, DECODE_PMC_FILE)))
        if d.get("kernel_source_sha256_16") != source_hash(DECODE_SOURCES):
            return {}
        return d.get("hbm_bytes_per_step", {})
    except Exception:
        return {}


def cpu_reference_object(dims, B=4):
    """SURVEY §8(d)'s CPU leg as specified: the object evaluation.py:72-86 builds -- stock `transformers` Whisper (third
    party, in the image; eager attention, fp32) with the build's own 3-line conv stack installed -- on the host cores.
    Train: B=4, 1 warm-up + 3 timed forward+backward passes, trainables = the conv stem (the adapters' extra weight
    gradients, ~6 % of the FLOPs, are absent: peft is not in the image).  Decode: B=4, 40 new tokens, greedy and beam-5 +
    repetition penalty 5 + no-repeat-2 through GenerationMixin.generate (the reference's call shape,
    evaluation.py:369-386)."""
    import torch
    import transformers
    from neuspeech1_amd.weights import synth_batch
    from tools.hf_reference_object import build_reference_object, generate, generate_kwargs
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    model = build_reference_object(dims, "cpu")
    x, labels = synth_batch(dims, B, 1234)
    xt, lt = torch.from_numpy(x), torch.from_numpy(labels)

    def step():
        model.zero_grad(set_to_none=True)
        model(input_features=xt, labels=lt).loss.backward()
    model.train()
    step()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    train = B * 3 / (time.perf_counter() - t0)
    model.eval()
    out = {"value": round(train, 4), "unit": "samples/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "reference",
           "kind_detail": "reference object: stock transformers Whisper + the conv stack, i.e. what /root/reference/evaluation.py:72-86 "
                          "builds (the reference's own files cannot travel to the GPU box; peft is not in the image)",
           "unit_decode": "tokens/s",
           "sample": f"stock transformers {transformers.__version__} Whisper (eager, fp32) + conv stack, whisper-base {dims.ch}-ch, "
                     f"B={B}: 1+3 fwd+bwd passes; GenerationMixin.generate 40 new tokens"}
    with torch.no_grad():
        for name, kw in (("greedy", dict(num_beams=1)), ("beam5_rep5_ngram2", dict(num_beams=5, repetition_penalty=5.0, no_repeat_ngram_size=2))):
            t0 = time.perf_counter()
            o = generate(model, xt, **kw, **generate_kwargs(dims, lt[:, :4].clone(), 40, suppress_eos=True))
            out[name] = round(B * (o.shape[1] - 4) / (time.perf_counter() - t0), 2)
    return out


def gpu_reference_object(dev, B_train=64, B_dec=128, new=64):
    """Context, never `value`: the SAME reference object on this GPU through stock PyTorch-ROCm under
    torch.autocast('cuda', fp16) -- the reference's own numerics and call shapes (finetune.py:242, evaluation.py:350,
    369-386) -- i.e. what a user of the reference gets on an MI355X without this library.  Train: whisper-base 208-ch,
    B=64, loss scaled by 65536, 1 warm-up + 3 timed forward+backward passes (trainables = the conv stem; no adapters:
    peft is not in the image).  Decode: whisper-base 273-ch, B=128, 64 new tokens with EOS suppressed, greedy and
    beam-5 + repetition penalty 5 + no-repeat-2."""
    import torch
    import transformers
    from neuspeech1_amd.weights import WhisperDims, synth_batch
    from tools.hf_reference_object import build_reference_object, generate, generate_kwargs
    out = {"kind": "reference object through stock PyTorch-ROCm, fp16 autocast, eager attention", "torch": torch.__version__,
           "transformers": transformers.__version__}
    dims = WhisperDims()
    model = build_reference_object(dims, dev)
    model.train()
    x, labels = synth_batch(dims, B_train, 1234)
    xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)

    def step():
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = model(input_features=xt, labels=lt).loss
        (loss * 65536.0).backward()
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    out["train_samples_per_s"] = round(B_train * 3 / (time.perf_counter() - t0), 2)
    out["train_sample"] = f"whisper-base 208-ch, B={B_train}, 1+3 fwd+bwd passes"
    del model, xt, lt
    torch.cuda.empty_cache()
    dims = WhisperDims(ch=273)
    model = build_reference_object(dims, dev, train_convs=False)
    x, labels = synth_batch(dims, B_dec, 1234)
    xt, pr = torch.from_numpy(x).to(dev), torch.from_numpy(labels[:, :4].copy()).to(dev)
    out["unit_decode"] = "tokens/s"
    out["decode_sample"] = f"whisper-base 273-ch, B={B_dec}, prompt 4, {new} new tokens, EOS suppressed, encoder pass included"
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for name, kw in (("greedy", dict(num_beams=1)), ("beam5_rep5_ngram2", dict(num_beams=5, repetition_penalty=5.0, no_repeat_ngram_size=2))):
            best = None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o = generate(model, xt, **kw, **generate_kwargs(dims, pr.clone(), new, suppress_eos=True))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[name] = round(B_dec * (o.shape[1] - 4) / best, 1)
    del model
    torch.cuda.empty_cache()
    return out


def decode_bytes_per_step(dims, B, nb, t):
    """algorithmic HBM bytes of ONE decode step at sequence position t: the decoder's fp16 weights once
    (per layer: self q|k|v + out, cross q + out, fc1, fc2; plus the tied LM head), the cross-attention K and V of every
    SEQUENCE once (shared by its beams), the self-attention K/V cache of every beam row up to t."""
    d, f, nl = dims.d, dims.ffn, dims.dec_layers
    w = nl * (4 * d * d + 2 * d * d + 2 * d * f) * 2 + dims.vocab * d * 2
    cross = B * nl * 2 * dims.src_pos * d * 2
    self_kv = B * nb * nl * 2 * t * d * 2
    return w + cross + self_kv


def eval_tokens_per_s(dev):
    """BASELINE's second metric (eval tokens/sec): whisper-base, 273-ch, B = 128, 64 new tokens with
    EOS suppressed so every row does identical work; tokens/s = emitted tokens (prompt excluded) / wall time including
    the encoder pass.  Reported beside the headline metric, never as `value`.  `roofline` = the HBM roofline of the
    decode STEP: algorithmic bytes per step / measured time per step (difference of a 64- and a 32-token run, so the
    encoder pass and the prompt cancel), against the 8 TB/s spec and the 6.3 TB/s achievable rate."""
    import torch
    from neuspeech1_amd.engine import MegWhisperEngine
    from neuspeech1_amd.generate import Generator
    from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch
    dims = WhisperDims(ch=273)
    eng = MegWhisperEngine(dims, make_state_dict(dims, 42), device=dev)
    gen = Generator(eng)
    B, NEW = 128, 64
    x, labels = synth_batch(dims, B, 1234)
    x = torch.from_numpy(x).to(dev)
    prompt = torch.from_numpy(labels[:, :4].copy()).to(dev)
    out = {"workload": "whisper-base 273-ch, B=128, prompt 4, 64 new tokens, EOS suppressed, encoder pass included", "unit": "tokens/s",
           "roofline": {}}

    def run(nb, kw, new):
        best = first = None
        for _ in range(3):      # (an evaluation run calls generate() once per batch with one signature: the third call is its steady state --
            torch.cuda.synchronize()   # the first records launch lists, the second captures the session's hipGraphs, later ones replay them)
            t0 = time.perf_counter()
            o = gen.generate(x, prompt, num_beams=nb, max_new_tokens=new, suppress_tokens=[dims.eos_id], check_every=8, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            first = dt if first is None else first
            best = dt if best is None else min(best, dt)
        return best, o.shape[1] - 4, gen.last_loop_mode, first

    for name, nb, kw in (("greedy", 1, {}), ("beam5_rep5_ngram2", 5, dict(repetition_penalty=5.0, no_repeat_ngram_size=2))):
        t64, n64, mode, first64 = run(nb, kw, NEW)
        t32, n32, _, _ = run(nb, kw, NEW // 2)
        out[name] = round(B * n64 / t64, 1)
        # the FIRST call of a signature (records the launch lists; nothing captured yet), beside the steady state `value` quotes
        out.setdefault("first_call_ms", {})[name] = round(first64 * 1e3, 2)
        out.setdefault("steady_call_ms", {})[name] = round(t64 * 1e3, 2)
        out.setdefault("loop_mode", {})[name] = mode      # "lists", or "lists->graphs@<step>" where the host turned out to be the bottleneck
        ms_step = (t64 - t32) / max(n64 - n32, 1) * 1e3
        t_mid = 4 + (n32 + n64) // 2                       # mean cache length over the differenced steps
        by = decode_bytes_per_step(dims, B, nb, t_mid)
        ach = by / (ms_step * 1e-3) / 1e9
        tr = _decode_pmc_traffic().get(name)
        out["roofline"][name] = {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                                 "frac_of_achievable_6300": round(ach / 6300.0, 4), "ms_per_step": round(ms_step, 4),
                                 "algorithmic_bytes_per_step": by, "traffic": round(tr) if tr else None,
                                 "traffic_source": DECODE_PMC_FILE}
    return out


def large_v2_leg(dev, B=64, steps=3):
    """BASELINE configs[4] on the one GPU of this run: whisper-large-v2 (32 + 32 layers, d 1280), 273-ch, fp16 LoRA r = 32
    (dropout 0.05) + conv-stem training step at SURVEY 8(d)'s B = 64 per GPU (169 GB of the 288 GB; rounds 1-4 ran B = 32):
    1 warm-up (eager) + 1 capture + `steps` timed graph replays.  Reported beside the headline, never as `value`."""
    import torch
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    from neuspeech1_amd.weights import WHISPER_LARGE_V2, make_state_dict, synth_batch
    dims = WHISPER_LARGE_V2
    t0 = time.perf_counter()
    rng = torch.Generator(device=dev)
    rng.manual_seed(42)

    def gen(name, shape, std, seed, mean=0.0):      # random init of the architecture, drawn on the device
        return torch.randn(tuple(shape) if not isinstance(shape, int) else (shape,), device=dev, generator=rng) * std + mean
    sd = make_state_dict(dims, 42, gen=gen)
    torch.manual_seed(42)
    eng = MegWhisperEngine(dims, sd, lora=LoraSpec(r=32, alpha=64.0, dropout=0.05),
                           train_cfg=TrainCfg(lr=1e-4, warmup_steps=0, total_steps=0), device=dev)
    del sd
    x, labels = synth_batch(dims, B, 1234)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    for _ in range(2):
        eng.train_step(xd, ld)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = eng.train_step(xd, ld)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"workload": f"whisper-large-v2 273-ch MEG (B,273,6000), fp16 LoRA r=32 (dropout 0.05) + conv-stem training step, bs{B}, "
                       f"label len {labels.shape[1]}", "value": round(B / dt, 2), "unit": "samples/s", "ms_per_step": round(dt * 1e3, 2),
           "steps": steps, "warmup": 2, "algorithmic_gflop_per_sample": LV2_GFLOP_PER_SAMPLE,
           "whole_step_mfma_frac": round(B / dt * LV2_GFLOP_PER_SAMPLE * 1e9 / (MFMA_PEAK_TFLOPS * 1e12), 4),
           "final_loss": round(float(loss.item()), 4), "graph": eng.graph_stats(), "setup_s": round(t_setup, 1),
           "mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    del eng
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE: 64)")
    ap.add_argument("--ch", type=int, default=208)
    ap.add_argument("--lora-r", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-eval", action="store_true", help="skip the decode (eval tokens/s) leg")
    ap.add_argument("--no-large-v2", action="store_true", help="skip the whisper-large-v2 leg (BASELINE configs[4] at N = 1)")
    args = ap.parse_args()

    from neuspeech1_amd import build as _b
    if not os.path.exists(_b.LIB):      # the prebuilt in-tree .so normally travels with the tree (no CPU fallback exists)
        if int(os.environ.get("RANK", "0")) == 0:
            _b.build()
        else:
            t_end = time.time() + 600   # rank 0 links to a temporary name and renames: the file appears complete
            while not os.path.exists(_b.LIB):
                if time.time() > t_end:
                    raise RuntimeError(f"rank {os.environ.get('RANK')}: {_b.LIB} did not appear within 600 s")
                time.sleep(1.0)
    import torch
    import torch.distributed as dist
    from neuspeech1_amd import ops
    from neuspeech1_amd.dp import GradReducer
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    from neuspeech1_amd.weights import WhisperDims, make_state_dict, synth_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("NS_DIST_BACKEND", "nccl") != "nccl":
        local %= max(1, torch.cuda.device_count())      # test mode: ranks may share a device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI; NS_DIST_BACKEND=gloo exists for the two-ranks-on-one-GPU test (RCCL refuses a shared device)
        if os.environ.get("NS_DIST_BACKEND", "nccl") == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(os.environ["NS_DIST_BACKEND"])

    dims = WhisperDims(ch=args.ch)
    B = args.batch
    sd = make_state_dict(dims, 42)          # identical seeded init on every rank (no broadcast needed)
    spec = LoraSpec(r=args.lora_r, alpha=2.0 * args.lora_r, dropout=0.05)
    torch.manual_seed(42)                   # the adapters' kaiming init must be the same replica on every rank
    eng = MegWhisperEngine(dims, sd, lora=spec, lora_sd=None,
                           train_cfg=TrainCfg(lr=1e-3, warmup_steps=500, total_steps=100000), device=dev)
    del sd
    x, labels = synth_batch(dims, B, 1234 + rank)    # disjoint shard per rank
    xd = torch.from_numpy(x).to(dev)
    ld = torch.from_numpy(labels).to(dev)
    red = GradReducer(eng.G, timing=True) if world > 1 else None

    def step():
        if red is None:
            return eng.train_step(xd, ld)
        return eng.train_step(xd, ld, on_ready=red.on_ready, reduce_fn=red.finish)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enq = time.perf_counter() - t0     # host time to ENQUEUE the timed steps (hipGraph replays; nothing waits for the GPU)
    if world > 1:
        torch.cuda.synchronize()
        t_local = time.perf_counter()       # this rank's own finish, before the barrier levels the ranks
        dist.barrier()
    torch.cuda.synchronize()
    if world == 1:
        t_local = time.perf_counter()
    dt = time.perf_counter() - t0
    dt_local = t_local - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_v = float(loss.item())
    value = world * B * args.steps / dt
    # what the timed loop actually ran on, read AFTER it: a capture that failed and fell back to eager steps shows here
    gstat = eng.graph_stats()
    gstat["timed_steps_replayed"] = gstat["replays"] >= args.steps and gstat["capture_failures"] == 0
    if world > 1:
        flags = torch.tensor([gstat["replays"], gstat["capture_failures"], gstat["captures"]], device=dev, dtype=torch.int64)
        allf = [torch.zeros_like(flags) for _ in range(world)]
        dist.all_gather(allf, flags)
        gstat["per_rank"] = [{"replays": int(f[0]), "capture_failures": int(f[1]), "captures": int(f[2])} for f in allf]
        gstat["timed_steps_replayed"] = all(int(f[0]) >= args.steps and int(f[1]) == 0 for f in allf)

    roof = budget = None
    if rank == 0 and not args.no_roofline:
        # dominant kernel = the 256x256 LDS-DMA ring GEMM: one extra instrumented step, HIP events on the launch stream
        ops.GEMM_PROFILE = []
        eng.train_step(xd, ld)      # local step, NO collective: only rank 0 runs this instrumented extra step
        torch.cuda.synchronize()
        recs, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
        tot = {}
        for kind, fl, e0, e1, by in recs:
            a = tot.setdefault(kind, [0.0, 0.0, 0, 0.0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
            a[3] += by
        dom = "nt256" if "nt256" in tot else "nt128"
        fl, sec, n, by = tot[dom]
        ach = fl / sec / 1e12
        sustained = sustained_mfma_tflops()
        traffic = _pmc_traffic(DOMINANT if dom == "nt256" else None)
        roof = {"bound": "mfma", "kernel": DOMINANT if dom == "nt256" else "ns_gemm_ring_kernel", "achieved": round(ach, 2),
                "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                "sustained_peak": sustained, "frac_of_sustained": round(ach / sustained, 4) if sustained else None,
                "sustained_source": YARDSTICK_FILE + " (back-to-back v_mfma_f32_16x16x32_f16 on random operands, every CU, seconds)",
                "traffic": traffic, "traffic_source": PMC_FILE,
                "algorithmic_bytes_per_launch": round(by / n),
                "traffic_over_algorithmic": round(traffic / (by / n), 3) if traffic else None,
                "launches_per_step": n, "avg_launch_ms": round(sec / n * 1e3, 4),
                "gflop_per_launch": round(fl / n / 1e9, 2),
                "step_share": {k: {"ms": round(v[1] * 1e3, 3), "tflops": round(v[0] / max(v[1], 1e-12) / 1e12, 1),
                                   "launches": v[2]} for k, v in tot.items()}}
        # the step's byte / FLOP budget by kernel class (one more instrumented eager step)
        ops.STEP_PROFILE = []
        eng.train_step(xd, ld)
        torch.cuda.synchronize()
        srecs, ops.STEP_PROFILE = ops.STEP_PROFILE, None
        budget = step_budget(srecs, dt / args.steps * 1e3, sustained)
    # encoder-only forward + backward (the quantity north_star's 40 % MFMA target is stated on): HIP events on the
    # launch stream at the encoder's section boundaries, one extra local step on rank 0
    enc = None
    if rank == 0 and not args.no_roofline:
        eng.section_events = {}
        eng.train_step(xd, ld)
        torch.cuda.synchronize()
        ev_, eng.section_events = eng.section_events, None
        fwd_ms = ev_["enc_fwd_begin"].elapsed_time(ev_["enc_fwd_end"])
        bwd_ms = ev_["enc_bwd_begin"].elapsed_time(ev_["enc_bwd_end"])
        gfe = ENC_GFLOP_PER_SAMPLE.get(args.ch, ENC_GFLOP_PER_SAMPLE[208])
        enc = {"fwd_ms": round(fwd_ms, 3), "bwd_ms": round(bwd_ms, 3), "algorithmic_gflop_per_sample": gfe,
               "mfma_frac": round(B * gfe * 1e9 / ((fwd_ms + bwd_ms) * 1e-3) / (MFMA_PEAK_TFLOPS * 1e12), 4)}
        sus = sustained_mfma_tflops()
        enc["mfma_frac_of_sustained"] = round(B * gfe * 1e9 / ((fwd_ms + bwd_ms) * 1e-3) / (sus * 1e12), 4) if sus else None

    dp = None
    if red is not None:
        torch.cuda.synchronize()
        ms_, by_ = red.exposed_ms(last=args.steps)
        # every rank's own wall time over the timed steps: the skew says whether one rank (host, clocks, a slow GPU) holds
        # the others at the collective
        mine = torch.tensor([dt_local / args.steps * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [round(float(t.item()), 3) for t in allr]
        dp = {"allreduce_bytes_per_step": by_, "exposed_allreduce_ms_per_step": round(ms_, 4),
              "allreduce_ms_per_step": round(red.total_ms(last=args.steps), 4), "chunks": 3,
              "rank_ms_per_step": per_rank, "rank_skew_ms": round(max(per_rank) - min(per_rank), 3),
              "backend": os.environ.get("NS_DIST_BACKEND", "nccl")}

    cpu = cpu_port = gpu_ref = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # `cpu_baseline` = SURVEY 8(d)'s leg (the reference object, B = 4, on the host cores); the oracle port beside it
        cpu_port = cpu_baseline(dims, args.lora_r, 2.0 * args.lora_r)
        try:
            cpu = cpu_reference_object(dims)
        except Exception as e:      # a reported baseline, not the product: never fail the bench line over it
            cpu = dict(cpu_port, note="reference-object leg failed: " + repr(e)[:160])

    ev = lv2 = None
    if rank == 0 and world == 1 and not args.no_eval:
        del eng
        torch.cuda.empty_cache()
        ev = eval_tokens_per_s(dev)
        if not args.no_large_v2:
            try:
                lv2 = large_v2_leg(dev)
            except Exception as e:      # a side leg: never fail the bench line over it
                lv2 = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_eval:
        try:
            gpu_ref = gpu_reference_object(dev)
        except Exception as e:
            gpu_ref = {"error": repr(e)[:200]}

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        gf = GFLOP_PER_SAMPLE.get(args.ch, 239.67)
        out = {
            "metric": "train samples/sec (whisper-base, 208-ch, bs64/GPU)", "value": round(value, 2),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
            "config": {"workload": f"whisper-base {args.ch}-ch MEG (B,{args.ch},6000), fp16 LoRA r={args.lora_r} "
                                   f"(dropout 0.05) + conv-stem training step, bs{B}/GPU, label len {labels.shape[1]}",
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "algorithmic_gflop_per_sample": gf,
                       "whole_step_mfma_frac": round(value / world * gf * 1e9 / (MFMA_PEAK_TFLOPS * 1e12), 4),
                       "encoder_fwd_bwd_mfma_frac": enc["mfma_frac"] if enc else None,
                       "final_loss": round(loss_v, 4),
                       "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
                       "train_step_graph": bool(gstat["timed_steps_replayed"]), "graph": gstat},
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_port": cpu_port, "torch_rocm_reference_object": gpu_ref,
            "encoder_fwd_bwd": enc, "dp": dp, "eval": ev, "large_v2": lv2, "step_budget": budget,
        }
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
