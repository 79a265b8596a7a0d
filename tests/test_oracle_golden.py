"""CPU: pin the oracle (oracle/whisper_meg_oracle.py) against golden vectors produced by the reference
object (stock HF Whisper + the reference's projection_module; tools/make_goldens.py)."""
import os

import numpy as np
import pytest
import torch

from neuspeech1_amd.weights import TINY, WHISPER_BASE, make_lora_state, make_state_dict, synth_batch
from oracle import whisper_meg_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
torch.set_num_threads(8)


def test_oracle_matches_reference_tiny_fwd_bwd():
    g = np.load(os.path.join(G, "train_tiny.npz"))
    sd = make_state_dict(TINY, int(g["seed_w"]))
    x, labels = synth_batch(TINY, int(g["B"]), int(g["seed_d"]))
    assert np.array_equal(labels, g["labels"])
    loss, logits, enc, grads = O.loss_and_grads(sd, None, x, labels, TINY, 0.0)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    np.testing.assert_allclose(enc.numpy(), g["enc"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=3e-4, rtol=1e-4)
    for k in (k for k in O.TRAINABLE_CONV if k in grads):
        got = grads[k].numpy()
        if ("grad." + k) in g:
            np.testing.assert_allclose(got, g["grad." + k], atol=2e-5, rtol=2e-3)
        else:
            np.testing.assert_allclose(got[:48, :48], g["gradblock." + k], atol=2e-5, rtol=2e-3)
            assert abs(np.sqrt((got.astype(np.float64) ** 2).sum()) - float(g["gradnorm." + k])) < 1e-4 * float(g["gradnorm." + k])


def test_oracle_lora_equals_reference_on_merged_weights():
    g = np.load(os.path.join(G, "lora_merged_tiny.npz"))
    r, alpha = int(g["r"]), float(g["alpha"])
    sd = make_state_dict(TINY, 42)
    lora = make_lora_state(TINY, r)
    x, labels = synth_batch(TINY, int(g["B"]), 1234)
    with torch.no_grad():
        loss, logits, _ = O.forward(O.to_torch(sd), torch.from_numpy(x), TINY, labels=torch.from_numpy(labels),
                                    lora=O.to_torch(lora), scale=alpha / r)
    assert abs(loss.item() - float(g["loss"])) < 5e-5
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=5e-4, rtol=1e-4)


def test_oracle_full_model_lora_equals_reference_on_merged_weights():
    """--ft_full: adapters on the decoder projections too, pinned the same way (tools/make_goldens.py lora_full)"""
    g = np.load(os.path.join(G, "lora_merged_tiny_full.npz"))
    r, alpha = int(g["r"]), float(g["alpha"])
    sd = make_state_dict(TINY, 42)
    lora = make_lora_state(TINY, r, decoder=True)
    x, labels = synth_batch(TINY, int(g["B"]), 1234)
    with torch.no_grad():
        loss, logits, _ = O.forward(O.to_torch(sd), torch.from_numpy(x), TINY, labels=torch.from_numpy(labels),
                                    lora=O.to_torch(lora), scale=alpha / r)
        loss_enc_only, _, _ = O.forward(O.to_torch(sd), torch.from_numpy(x), TINY, labels=torch.from_numpy(labels),
                                        lora=O.to_torch(make_lora_state(TINY, r)), scale=alpha / r)
    assert abs(loss.item() - float(g["loss"])) < 5e-5
    assert abs(loss_enc_only.item() - float(g["loss"])) > 1e-3       # the decoder adapters matter in this fixture
    np.testing.assert_allclose(logits.numpy()[:, :, ::7], g["logits"], atol=5e-4, rtol=1e-4)


@pytest.mark.parametrize("tag,ch,B", [("base208", 208, 2)])
def test_oracle_matches_reference_base_shape(tag, ch, B):
    g = np.load(os.path.join(G, f"train_{tag}.npz"))
    dims = WHISPER_BASE
    sd = make_state_dict(dims, 42)
    x, labels = synth_batch(dims, B, 1234)
    loss, logits, enc, grads = O.loss_and_grads(sd, None, x, labels, dims, 0.0)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    np.testing.assert_allclose(enc.numpy()[:, ::97, :16], g["enc_slice"], atol=5e-4, rtol=1e-3)
    np.testing.assert_allclose(logits.numpy()[:, :, :16], g["logits_slice"], atol=2e-3, rtol=1e-3)
    assert np.array_equal(logits.numpy().argmax(-1)[g["top_margin"] > 1e-3], g["top1_id"][g["top_margin"] > 1e-3])
    for k in (k for k in O.TRAINABLE_CONV if k in grads):
        gr = grads[k].numpy()
        n = np.sqrt((gr.astype(np.float64) ** 2).sum())
        assert abs(n - float(g["gradnorm." + k])) < 2e-3 * float(g["gradnorm." + k]), k


def test_reader_collate_and_matchers():
    rng = np.random.default_rng(0)
    for n_ch, name, modal_ch, lo in ((224, "gwilliams", 208, 0), (301, "schoffelen", 273, 28), (100, None, 208, 0)):
        for n in (700, 6000, 7321):
            s = rng.standard_normal((n_ch, n))
            out = O.reader_pad_sample(s, name, modal_ch, 6000)
            assert out.shape == (modal_ch, 6000) and out.dtype == np.float64
            m = min(n, 6000)
            keep = min(n_ch - lo, modal_ch)
            np.testing.assert_array_equal(out[:keep, :m], s[lo:lo + keep, :m])
            assert not out[:, m:].any() and not out[keep:].any()
    feats = [{"input_features": [np.ones((3, 8))], "labels": [7, 1, 2]}, {"input_features": [np.zeros((3, 8))], "labels": [7, 5]}]
    b = O.collate(feats, pad_id=9, bos_id=7)
    assert b["input_features"].dtype == torch.float32 and b["input_features"].shape == (2, 3, 8)
    assert b["labels"].tolist() == [[1, 2], [5, -100]]
    b = O.collate(feats, pad_id=9, bos_id=3)
    assert b["labels"].tolist() == [[7, 1, 2], [7, 5, -100]]
    names = [f"model.encoder.layers.{i}.{m}" for i in range(6) for m in
             ("self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj", "self_attn.out_proj", "self_attn_layer_norm",
              "fc1", "fc2", "final_layer_norm")] + ["model.decoder.layers.0.fc1", "model.encoder.conv1"]
    got = O.match_modules_string(names, ["model.encoder"], ["k_proj", "q_proj", "v_proj", "out_proj", "fc1", "fc2"])
    assert len(got) == 36 and all(n.startswith("model.encoder.layers.") for n in got)
    assert O.shift_tokens_right(torch.tensor([[5, 6, -100]]), 9, 1).tolist() == [[1, 5, 6]]


def test_oracle_decode_matches_reference_generate():
    """greedy + beam-5 (with/without repetition penalty 5.0 + no-repeat-2, with EOS finishing) vs HF GenerationMixin."""
    g = np.load(os.path.join(G, "decode_tiny.npz"))
    dims = TINY
    sd = O.to_torch(make_state_dict(dims, 42))
    x, labels = synth_batch(dims, int(g["B"]), 1234)
    prompt = torch.from_numpy(labels[:, :4].copy())
    xt = torch.from_numpy(x)
    n = int(g["new_tokens"])

    def same(name, a):
        ref, a = g[name], a.numpy()
        Lm = min(a.shape[1], ref.shape[1])
        assert np.array_equal(a[:, :Lm], ref[:, :Lm]), name
        assert (a[:, Lm:] == dims.pad_id).all() and (ref[:, Lm:] == dims.pad_id).all(), name

    rp = dict(repetition_penalty=5.0, no_repeat_ngram_size=2)
    with torch.no_grad():
        same("greedy", O.greedy(sd, xt, dims, prompt, n))
        same("greedy_rp", O.greedy(sd, xt, dims, prompt, n, **rp))
        same("beam5", O.beam_search(sd, xt, dims, prompt, 5, n))
        same("beam5_rp", O.beam_search(sd, xt, dims, prompt, 5, n, **rp))
        for eos in (34, 630):
            same(f"greedy_eos{eos}", O.greedy(sd, xt, dims, prompt, n, eos_id=eos))
            same(f"beam5_eos{eos}", O.beam_search(sd, xt, dims, prompt, 5, n, eos_id=eos))
            same(f"beam5_rp_eos{eos}", O.beam_search(sd, xt, dims, prompt, 5, n, eos_id=eos, **rp))


def test_oracle_adalora_equals_reference_on_merged_weights():
    """AdaLoRA -- the reference's default adapter (finetune.py:43,205-208) -- pinned like LoRA: the oracle's
    y += B((A x) * E) * alpha/(r + 1e-5) against the reference object run on W + alpha/(r+1e-5) B (A * E)
    (tools/make_goldens.py adalora), with a non-zero E."""
    g = np.load(os.path.join(G, "adalora_merged_tiny.npz"))
    r, alpha = int(g["r"]), float(g["alpha"])
    sd = make_state_dict(TINY, 42)
    lora = make_lora_state(TINY, r, adalora=True, b_std=float(g["b_std"]))
    x, labels = synth_batch(TINY, int(g["B"]), 1234)
    with torch.no_grad():
        loss, logits, _ = O.forward(O.to_torch(sd), torch.from_numpy(x), TINY, labels=torch.from_numpy(labels),
                                    lora=O.to_torch(lora), scale=alpha / (r + 1e-5))
    assert abs(loss.item() - float(g["loss"])) < 5e-5
    assert abs(float(g["loss"]) - float(g["loss_base"])) > 1e-2          # the adapter matters in this fixture
    np.testing.assert_allclose(logits.numpy()[:, :, ::3], g["logits"], atol=5e-4, rtol=1e-4)
    # and the merge itself (what merge_and_unload does) reproduces the same function
    merged = O.lora_merge(sd, lora, alpha / (r + 1e-5))
    with torch.no_grad():
        loss_m, _, _ = O.forward(O.to_torch(merged), torch.from_numpy(x), TINY, labels=torch.from_numpy(labels))
    assert abs(loss_m.item() - float(g["loss"])) < 5e-5


@pytest.mark.parametrize("tag", ["base273", "lv2w"])
def test_oracle_matches_reference_273ch_and_large_v2_width(tag):
    """BASELINE configs[3] / [4] shapes: whisper-base with 273 channels, and whisper-large-v2's WIDTH (d 1280, 20 heads,
    ffn 5120) at 2 + 2 layers, B = 1."""
    from neuspeech1_amd.weights import LV2W, WhisperDims
    g = np.load(os.path.join(G, f"train_{tag}.npz"))
    dims = LV2W if tag == "lv2w" else WhisperDims(ch=273)
    sd = make_state_dict(dims, 42)
    x, labels = synth_batch(dims, int(g["B"]), 1234)
    assert np.array_equal(labels, g["labels"])
    loss, logits, enc, grads = O.loss_and_grads(sd, None, x, labels, dims, 0.0)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    np.testing.assert_allclose(enc.numpy()[:, ::97, :16], g["enc_slice"], atol=5e-4, rtol=1e-3)
    np.testing.assert_allclose(logits.numpy()[:, :, :16], g["logits_slice"], atol=2e-3, rtol=1e-3)
    for k in (k for k in O.TRAINABLE_CONV if k in grads):
        n = np.sqrt((grads[k].numpy().astype(np.float64) ** 2).sum())
        assert abs(n - float(g["gradnorm." + k])) < 2e-3 * float(g["gradnorm." + k]), k


def test_oracle_forced_decoder_ids_match_reference_generate():
    """forced_decoder_ids + non-empty suppress / begin-suppress lists (what a hub whisper generation_config.json
    carries): greedy and beam-5 ids of the reference object (tools/make_goldens.py forced), prompt lengths 1 and 4."""
    g = np.load(os.path.join(G, "decode_tiny_forced.npz"))
    dims = TINY
    sd = O.to_torch(make_state_dict(dims, 42))
    x, _ = synth_batch(dims, int(g["B"]), 1234)
    xt = torch.from_numpy(x)
    n = int(g["new_tokens"])
    forced = [[int(i), None if int(t) < 0 else int(t)] for i, t in zip(g["forced_idx"], g["forced_tok"])]
    kw = dict(suppress_tokens=g["suppress"].tolist(), begin_suppress_tokens=g["begin_suppress"].tolist(),
              forced_decoder_ids=forced)
    rp = dict(repetition_penalty=5.0, no_repeat_ngram_size=2)
    with torch.no_grad():
        for pn in ("p1", "p4"):
            prompt = torch.from_numpy(g[pn + ".prompt"])
            assert np.array_equal(O.greedy(sd, xt, dims, prompt, n, **kw).numpy(), g[pn + ".greedy"]), pn
            assert np.array_equal(O.greedy(sd, xt, dims, prompt, n, **rp, **kw).numpy(), g[pn + ".greedy_rp"]), pn
            assert np.array_equal(O.beam_search(sd, xt, dims, prompt, 5, n, **rp, **kw).numpy(), g[pn + ".beam5_rp"]), pn
            assert np.array_equal(O.beam_search(sd, xt, dims, prompt, 5, n, **kw).numpy(), g[pn + ".beam5"]), pn
    # the forced positions really are forced, and position 1 (None) is free
    assert (g["p1.greedy_rp"][:, 2] == forced[1][1]).all() and (g["p1.greedy_rp"][:, 3] == forced[2][1]).all()


@pytest.mark.parametrize("tag,src", [("tiny", "tiny"), ("tiny_sb", "tiny_sb"), ("tiny_forced", "tiny_forced"), ("base208", "base208"),
                                     ("base273", "base273"), ("lv2w", "lv2w")])
def test_reference_hypotheses_files_belong_to_their_goldens(tag, src):
    """decode_<tag>_hyps.npz (the reference object's top-num_beams finished hypotheses, tools/make_goldens.py hyps; VERDICT r4
    #3b): hypothesis 0 is the golden's own row, scores are sorted and hypothesis 0 carries the golden's score -- the GPU tests
    may then accept a beam row that left hypothesis 0 only if it IS one of these alternatives."""
    g = np.load(os.path.join(G, f"decode_{src}.npz"))
    h = np.load(os.path.join(G, f"decode_{tag}_hyps.npz"))
    names = [k[:-5] for k in h.files if k.endswith("_hyps")]
    assert names
    for n in names:
        hyps, sc = h[n + "_hyps"], h[n + "_hyp_scores"]
        ref = g[n]
        assert hyps.shape[:2] == sc.shape == (ref.shape[0], int(h["num_beams"]))
        L = min(ref.shape[1], hyps.shape[2])
        assert np.array_equal(hyps[:, 0, :L], ref[:, :L])
        assert (np.diff(sc, axis=1) <= 1e-6).all()
        if n + "_scores" in g.files:
            np.testing.assert_allclose(sc[:, 0], g[n + "_scores"], atol=1e-5)
        # distinct alternatives: no two hypotheses of a row are the same sequence
        for b in range(hyps.shape[0]):
            assert len({tuple(r) for r in hyps[b].tolist()}) == hyps.shape[1]


def test_oracle_sequence_score_reproduces_the_reference_hypothesis_scores():
    """O.sequence_score (what the GPU beam tests rescore a row with when it left the reference's best hypothesis) against the
    reference object's own `sequences_scores` of ALL its finished hypotheses: plain, with repetition penalty + no-repeat-2,
    with EOS-finished rows, and with a sequence-bias table."""
    dims = TINY
    sd = O.to_torch(make_state_dict(dims, 42))
    x, labels = synth_batch(dims, 3, 1234)
    xt = torch.from_numpy(x)
    rp = dict(repetition_penalty=5.0, no_repeat_ngram_size=2)
    h = np.load(os.path.join(G, "decode_tiny_hyps.npz"))
    hs = np.load(os.path.join(G, "decode_tiny_sb_hyps.npz"))
    sbg = np.load(os.path.join(G, "decode_tiny_sb.npz"))
    sb = {tuple(int(t) for t in str(k).split(",")): float(v) for k, v in zip(sbg["sequence_bias_keys"], sbg["sequence_bias_vals"])}
    cases = [(h, "beam5", {}), (h, "beam5_rp", rp), (h, "beam5_eos630", dict(eos_id=630)), (h, "beam5_rp_eos34", dict(eos_id=34, **rp)),
             (hs, "beam5_rp_sb", dict(sequence_bias=sb, **rp)), (hs, "beam5_sb", dict(sequence_bias=sb))]
    with torch.no_grad():
        for f, name, kw in cases:
            hyps, sc = f[name + "_hyps"], f[name + "_hyp_scores"]
            for b in range(hyps.shape[0]):
                for k in (0, 2, 4):
                    got = O.sequence_score(sd, xt[b:b + 1], dims, hyps[b, k], 4, **kw)
                    assert abs(got - float(sc[b, k])) < 2e-4, (name, b, k, got, float(sc[b, k]))
