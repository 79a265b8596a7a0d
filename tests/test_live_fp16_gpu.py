"""The authoritative numerics check of SURVEY.md §8c (G2) / Appendix A: the HIP path against a LIVE run of the
reference object on the GPU box under `torch.autocast('cuda', torch.float16)` -- the reference's own numerics
(/root/reference/evaluation.py:350, finetune.py:242) -- instead of the fp32-CPU goldens.

The reference object is stock `transformers` Whisper (third party, part of the image; eager attention, fp32 weights)
with the build's own conv stack installed (tools/hf_reference_object.py): no reference file is read here.

What is asserted:
  * Appendix A's dtype table, from module hooks under CUDA autocast (which ops round to fp16, fp32 residual stream);
  * logits and encoder states: HIP vs live-fp16 within a relative Frobenius error of 1.5e-3 (measured 6.3e-4 .. 7.0e-4;
    the fp32-golden bound of tests/test_engine_gpu.py is 1e-2), and the HIP path is no further from the object's fp32
    run than the live fp16 run is (factor 1.25; measured: HIP 6.2e-4 .. 7.8e-4, live fp16 6.5e-4 .. 8.1e-4);
  * loss within 2e-4 relative of the live fp16 loss (measured <= 4e-5);
  * greedy ids: equal to the live fp16 generation; a row may leave the live run only AT a position that NEITHER side
    decides (top-1 minus top-2 processed score <= one fp16 spacing in the live run and in this path's own trace) and
    only for the live run's runner-up token; at most one such row over three repetitions of a case.
"""
import os

import numpy as np
import pytest
import torch

from neuspeech1_amd.weights import TINY, WHISPER_BASE, WhisperDims, make_state_dict, synth_batch

pytestmark = pytest.mark.gpu

TIE = 0.02      # fp16 logits of magnitude 4-8 resolve 0.004-0.008: a decision closer than this is not determined at fp16
ONE_FP16_SPACING = 0.004      # 2^-8: the spacing of fp16 values in [4, 8)


def rel(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _obj(dims, dev):
    from tools.hf_reference_object import build_reference_object
    return build_reference_object(dims, dev)


def _engine(dims, dev):
    from neuspeech1_amd.engine import MegWhisperEngine
    return MegWhisperEngine(dims, make_state_dict(dims, 42), device=dev)


def test_autocast_dtype_table(dev):
    """SURVEY.md Appendix A, measured live: conv / linear outputs fp16, LayerNorm outputs fp32, the residual stream
    (layer outputs, encoder output) fp32, logits fp16, loss fp32."""
    dims = TINY
    model = _obj(dims, dev)
    seen = {}

    def hook(name):
        def f(mod, inp, out):
            o = out[0] if isinstance(out, tuple) else out
            if torch.is_tensor(o):
                seen[name] = o.dtype
        return f
    hs = [m.register_forward_hook(hook(n)) for n, m in model.named_modules() if n]
    x, labels = synth_batch(dims, 2, 5)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = model(input_features=torch.from_numpy(x).to(dev), labels=torch.from_numpy(labels).to(dev))
    for h in hs:
        h.remove()
    f16, f32 = torch.float16, torch.float32
    e, d_ = "model.encoder.", "model.decoder."
    expect = {
        e + "conv1.0": f16, e + "conv1.2": f16, e + "conv1": f16, e + "conv2": f16,
        e + "layers.0.self_attn_layer_norm": f32, e + "layers.0.self_attn.q_proj": f16, e + "layers.0.self_attn.k_proj": f16,
        e + "layers.0.self_attn.v_proj": f16, e + "layers.0.self_attn.out_proj": f16, e + "layers.0.self_attn": f16,
        e + "layers.0.final_layer_norm": f32, e + "layers.0.fc1": f16, e + "layers.0.fc2": f16,
        e + "layers.0": f32, e + "layers.1": f32, e + "layer_norm": f32,
        d_ + "embed_tokens": f32, d_ + "layers.0.self_attn_layer_norm": f32, d_ + "layers.0.self_attn.out_proj": f16,
        d_ + "layers.0.encoder_attn.q_proj": f16, d_ + "layers.0.encoder_attn.k_proj": f16, d_ + "layers.0.encoder_attn": f16,
        d_ + "layers.0.fc2": f16, d_ + "layers.0": f32, d_ + "layer_norm": f32, "proj_out": f16,
    }
    bad = {k: (seen.get(k), v) for k, v in expect.items() if seen.get(k) != v}
    assert not bad, bad
    assert out.logits.dtype == f16 and out.loss.dtype == f32
    assert out.encoder_last_hidden_state.dtype == f32


def _softmax_is_fp32_under_autocast(dev):
    with torch.autocast("cuda", dtype=torch.float16):
        s = torch.randn(4, 8, device=dev, dtype=torch.float16)
        return torch.softmax(s, -1).dtype, torch.nn.functional.layer_norm(s, (8,)).dtype, \
            torch.nn.functional.gelu(s).dtype, (s @ s.t()).dtype


def test_autocast_op_policy(dev):
    """the op-level policy Appendix A states: softmax and layer_norm -> fp32, gelu keeps its input dtype, matmul -> fp16"""
    sm, ln, ge, mm = _softmax_is_fp32_under_autocast(dev)
    assert sm == torch.float32 and ln == torch.float32 and ge == torch.float16 and mm == torch.float16


CASES = [("tiny", TINY, 3), ("base208", WHISPER_BASE, 2), ("base273", WhisperDims(ch=273), 2)]


@pytest.mark.parametrize("tag,dims,B", CASES, ids=[c[0] for c in CASES])
def test_logits_and_loss_vs_live_fp16(dev, tag, dims, B):
    model = _obj(dims, dev)
    eng = _engine(dims, dev)
    x, labels = synth_batch(dims, B, 1234)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.float16):
            live = model(input_features=xd, labels=ld)
        full = model(input_features=xd, labels=ld)          # the same object in fp32: the yardstick for both fp16 paths
    loss, logits = eng.forward(xd, ld, train=False)
    lg = logits.float()
    enc = eng._b["enc16"].float().view(B, dims.src_pos, dims.d)
    e_hip_live = rel(lg, live.logits.float())
    e_hip_full = rel(lg, full.logits)
    e_live_full = rel(live.logits.float(), full.logits)
    n_hip_live = rel(enc, live.encoder_last_hidden_state.float())
    n_hip_full = rel(enc, full.encoder_last_hidden_state)
    n_live_full = rel(live.encoder_last_hidden_state.float(), full.encoder_last_hidden_state)
    print(f"\n[{tag}] logits rel: hip-live {e_hip_live:.2e} hip-fp32 {e_hip_full:.2e} live-fp32 {e_live_full:.2e} | "
          f"enc rel: hip-live {n_hip_live:.2e} hip-fp32 {n_hip_full:.2e} live-fp32 {n_live_full:.2e} | "
          f"loss hip {loss.item():.5f} live {live.loss.item():.5f} fp32 {full.loss.item():.5f}")
    assert e_hip_live < 1.5e-3 and n_hip_live < 1.5e-3, (e_hip_live, n_hip_live)
    assert e_hip_full < 1.25 * e_live_full + 1e-4, (e_hip_full, e_live_full)
    assert n_hip_full < 1.25 * n_live_full + 1e-4, (n_hip_full, n_live_full)
    assert abs(loss.item() - live.loss.item()) < 2e-4 * live.loss.item()
    # top-1 ids of the teacher-forced pass equal the live run's wherever the live fp16 margin is decisive
    ll = live.logits.float()
    top2 = ll.topk(2, -1).values
    sure = (top2[..., 0] - top2[..., 1]) > TIE
    assert torch.equal(lg.argmax(-1)[sure], ll.argmax(-1)[sure])
    assert sure.float().mean().item() > 0.7      # flat random-init logits: 81-100 % of the positions are decisive


def _live_generate(model, dims, xd, prompt, new, **kw):
    from tools.hf_reference_object import generate, generate_kwargs
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        o = generate(model, xd, **generate_kwargs(dims, prompt.clone(), new), output_scores=True,
                     return_dict_in_generate=True, **kw)
    seq = o.sequences
    # processed scores per step -> the live run's own decision margin (greedy: row-wise top-1 minus top-2) and its two candidates
    top = [s.float().topk(2, -1) for s in o.scores]
    vals = torch.stack([t.values for t in top], 1)          # (rows, steps, 2)
    idx = torch.stack([t.indices for t in top], 1)
    return seq, (vals[..., 0] - vals[..., 1]), idx


REPEATS = 3


@pytest.mark.parametrize("tag,dims,B", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("name,kw", [("greedy", {}), ("greedy_rp", dict(repetition_penalty=5.0, no_repeat_ngram_size=2))])
def test_greedy_ids_vs_live_fp16(dev, tag, dims, B, name, kw):
    """north_star: greedy token ids bit-identical to the reference.  Against the reference's OWN numerics (a live fp16 run on this box)
    a row may leave it only at an undecided position, and VERDICT r4 #3(a) pins what "undecided" means from BOTH sides:
      * the live run's own top-1 minus top-2 processed score there is at most one fp16 spacing (0.0039 at magnitude 4-8);
      * the token this path chose IS the live run's runner-up of that step (not just any token);
      * this path's own top-1 minus top-2 there is at most one fp16 spacing too (it did not decide that position clearly either);
      * everything before the flip matches, and over REPEATS full repetitions of the case (the live side is not reproducible at a
        tie: identical live runs report different minimum margins) at most ONE row leaves in total."""
    from neuspeech1_amd.generate import Generator
    model = _obj(dims, dev)
    gen = Generator(_engine(dims, dev))
    B = max(B, 4)
    x, labels = synth_batch(dims, B, 1234)
    xd = torch.from_numpy(x).to(dev)
    prompt = torch.from_numpy(labels[:, :4].copy()).to(dev)
    new = 24
    P = prompt.shape[1]
    flips, min_margin = [], float("inf")
    first = None
    for rep in range(REPEATS):
        ref, margin, cand = _live_generate(model, dims, xd, prompt, new, num_beams=1, **kw)
        trace = []
        out = gen.generate(xd, prompt, num_beams=1, max_new_tokens=new, check_every=1, trace=trace, **kw)
        got, ref = out.cpu().numpy(), ref.cpu().numpy()
        if first is None:
            first = got
        assert np.array_equal(got, first), "the HIP generation itself must be reproducible run to run"
        min_margin = min(min_margin, float(margin.min()))
        Lm = min(got.shape[1], ref.shape[1])
        for b in range(B):
            neq = np.nonzero(got[b, :Lm] != ref[b, :Lm])[0]
            if len(neq) == 0:
                continue
            p = int(neq[0])
            assert p >= P, (tag, name, b, p)
            step = p - P
            m_live = float(margin[b, step])
            runner_up = int(cand[b, step, 1])
            hv, hi = trace[step]
            m_hip = float(hv[b, 0] - hv[b, 1])
            ctx = (tag, name, rep, b, p, m_live, m_hip, got[b].tolist(), ref[b].tolist())
            assert m_live <= ONE_FP16_SPACING, ctx                 # the live run did not decide this position
            assert int(got[b, p]) == runner_up, ctx                 # ... and this path took the live run's second candidate
            assert int(hi[b, 0]) == int(got[b, p]), ctx             # (the trace is of this very decision)
            assert m_hip <= ONE_FP16_SPACING, ctx                   # ... at a margin that is sub-spacing on this side too
            flips.append((rep, b, p, round(m_live, 4), round(m_hip, 4)))
    print(f"\n[{tag}/{name}] rows {B} x {REPEATS} repetitions, rows leaving the live fp16 run (rep, row, pos, live margin, hip margin): "
          f"{flips}; min live margin {min_margin:.4f}")
    assert len(flips) <= 1, flips
