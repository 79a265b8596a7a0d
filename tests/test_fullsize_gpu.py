"""BASELINE's FULL sizes, checked and not only timed (index overflow, grid-size and slab-workspace bugs live here):

  * configs[1]: one B = 64 whisper-base 208-ch training pass (LoRA r = 32 + conv stem).  The loss is a token mean and the
    gradient is linear in the per-sample terms, so loss and the flat gradient buffer G of the B = 64 pass must equal the
    token-weighted sum of eight B = 8 passes over the same rows -- a size-independent property that needs no oracle run
    at a size the CPU cannot finish.  (Each B = 8 pass is itself inside the size range the oracle / golden tests cover.)
  * configs[3]: B = 128, 273-ch, beam-5 + repetition penalty 5 + no-repeat-2 decode.  Rows 0, 1 are the inputs of the
    reference object's golden (tests/golden/decode_base273.npz) and must reproduce it inside the full batch; every row
    must decode to what the same input decodes to in a B = 2 batch.
"""
import os

import numpy as np
import pytest
import torch

from neuspeech1_amd.weights import WHISPER_BASE, WhisperDims, make_lora_state, make_state_dict, synth_batch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_b64_training_pass_equals_token_weighted_sum_of_eight_b8_passes(dev):
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    dims, r = WHISPER_BASE, 32
    eng = MegWhisperEngine(dims, make_state_dict(dims, 42), lora=LoraSpec(r=r, alpha=2.0 * r, dropout=0.0),
                           lora_sd=make_lora_state(dims, r), train_cfg=TrainCfg(), device=dev)
    x, labels = synth_batch(dims, 64, 1234)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)

    def one(xs, ls):
        eng.zero_grad()
        loss, _ = eng.forward(xs, ls, train=True, compute_grad=True)
        eng.backward()
        return loss.item(), eng.G.clone()
    loss64, G64 = one(xd, ld)
    assert np.isfinite(loss64) and torch.isfinite(G64).all()
    nv64 = int((labels != -100).sum())
    acc, loss_acc = torch.zeros_like(G64, dtype=torch.float64), 0.0
    for i in range(8):
        sl = slice(8 * i, 8 * i + 8)
        li, Gi = one(xd[sl].contiguous(), ld[sl].contiguous())
        w = int((labels[sl] != -100).sum()) / nv64
        acc += Gi.double() * w
        loss_acc += li * w
    assert abs(loss_acc - loss64) < 1e-3 * loss64, (loss_acc, loss64)
    assert rel(acc, G64) < 1e-3, rel(acc, G64)
    # per parameter tensor (a wrong row range or slab in ONE site would hide in the whole-buffer norm)
    worst = {}
    for name, (o, n) in eng.seg_off.items():
        if G64[o:o + n].abs().max() > 0:
            worst[name] = rel(acc[o:o + n], G64[o:o + n])
    bad = {k: v for k, v in worst.items() if not v < 4e-3}
    assert not bad, bad
    print(f"\nB=64 vs 8 x B=8: loss {loss64:.5f} vs {loss_acc:.5f}; G rel {rel(acc, G64):.2e}; worst tensor {max(worst.values()):.2e}")


@pytest.mark.parametrize("name,nb,kw", [("greedy_rp", 1, dict(repetition_penalty=5.0, no_repeat_ngram_size=2)),
                                        ("beam5_rp", 5, dict(repetition_penalty=5.0, no_repeat_ngram_size=2))])
def test_b128_273ch_decode_rows_equal_golden_and_small_batch_decodes(dev, name, nb, kw):
    from neuspeech1_amd.engine import MegWhisperEngine
    from neuspeech1_amd.generate import Generator
    g = np.load(os.path.join(G, "decode_base273.npz"))
    dims = WhisperDims(ch=273)
    gen = Generator(MegWhisperEngine(dims, make_state_dict(dims, 42), device=dev))
    new = int(g["new_tokens"])
    x2, l2 = synth_batch(dims, int(g["B"]), 1234)          # the golden's two inputs
    xr, lr = synth_batch(dims, 126, 4321, min_k=8, max_k=8)
    x = torch.from_numpy(np.concatenate([x2, xr])).to(dev)
    prompt = torch.from_numpy(np.concatenate([l2[:, :4], lr[:, :4]])).to(dev)
    out = gen.generate(x, prompt, num_beams=nb, max_new_tokens=new, check_every=1, **kw).cpu().numpy()
    scores = gen.last_scores.cpu().numpy().copy() if nb > 1 else None
    assert out.shape[0] == 128
    # rows 0, 1 inside the full batch against the reference object's ids (same allowance as the B = 2 golden test: a
    # greedy row may leave the reference only where the reference's own fp32 margin is below 0.03; beam rows carry the
    # reference's final score)
    ref = g[name]
    Lm = min(out.shape[1], ref.shape[1])
    for b in range(2):
        neq = np.nonzero(out[b, :Lm] != ref[b, :Lm])[0]
        if len(neq) and nb == 1:
            assert g[name + "_margin"][b, int(neq[0]) - 4] < 0.03, (b, out[b].tolist(), ref[b].tolist())
    if nb > 1:
        # beam rows 0 / 1: ids, under the rule of tests/test_generate_gpu.py::check_beam -- hypothesis 0 of the reference object, or
        # (at most one row) another of ITS OWN finished hypotheses within 2e-2 of its best (tests/golden/decode_base273_hyps.npz)
        np.testing.assert_allclose(scores[:2], g[name + "_scores"], atol=2e-2)
        hy = np.load(os.path.join(G, "decode_base273_hyps.npz"))
        hyps, hsc = hy[name + "_hyps"], hy[name + "_hyp_scores"]
        left = 0
        for b in range(2):
            def row_is(h):
                L = min(out.shape[1], len(h))
                return np.array_equal(out[b, :L], h[:L]) and (out[b, L:] == dims.pad_id).all() and (h[L:] == dims.pad_id).all()
            if row_is(hyps[b, 0]):
                continue
            alt = [k for k in range(1, hyps.shape[1]) if row_is(hyps[b, k])]
            if alt:
                assert hsc[b, 0] - hsc[b, alt[0]] < 2e-2, (b, out[b].tolist(), hyps[b].tolist(), hsc[b].tolist())
            else:       # outside the reference's final five: the reference's own arithmetic (oracle rescoring) must rate it a near-tie
                from tests.test_generate_gpu import make_rescore
                true = make_rescore(dims, torch.from_numpy(x2), 4, **kw)(b, out[b])
                assert abs(true - float(hsc[b, 0])) < 2e-2 and abs(true - float(scores[b])) < 2e-2, (b, true, scores[b], hsc[b].tolist())
            left += 1
        assert left <= 1
    # every row against the SAME input decoded in a batch of two
    diff, dscore = [], 0.0
    for i in range(0, 128, 2):
        o2 = gen.generate(x[i:i + 2].contiguous(), prompt[i:i + 2].contiguous(), num_beams=nb, max_new_tokens=new, check_every=1,
                          **kw).cpu().numpy()
        s2 = gen.last_scores.cpu().numpy() if nb > 1 else None
        L = min(o2.shape[1], out.shape[1])
        for j in range(2):
            same = np.array_equal(o2[j, :L], out[i + j, :L]) and (o2[j, L:] == dims.pad_id).all() and (out[i + j, L:] == dims.pad_id).all()
            if nb > 1:      # the hypothesis score must agree whether or not a near-tie moved the ids
                dscore = max(dscore, abs(float(s2[j]) - float(scores[i + j])))
            if not same:
                diff.append(i + j)
    print(f"\n[{name}] rows whose B=128 ids differ from their B=2 ids: {diff}; largest score difference {dscore:.4f}")
    assert dscore < 5e-2, dscore
    # the GEMM tile shapes differ between M = 640 and M = 10 rows, so fp32 summation order (and with it an fp16 rounding
    # here and there) may differ.  Greedy decoding has one decision per token and reproduces every row (measured: 0 of
    # 128 differ).  Beam-5 + penalties on a flat random-init model weighs 10 candidates per row and step and keeps
    # near-ties alive: measured 9 of 128 rows end on a different hypothesis whose length-normalised score equals the
    # B = 2 one within 0.024 (asserted above: < 0.05).  An index, grid-size or slab bug would break most rows and their scores.
    assert len(diff) <= (0 if nb == 1 else 11), diff      # measured 0 / 9 (+ 2 for box-to-box noise in the near-ties)
