"""Generate tests/golden/*.npz by running the REFERENCE OBJECT in the build container:
stock `transformers.WhisperForConditionalGeneration` (eager attention, fp32, CPU) with the
reference's own `utils/model_utils.projection_module('base', ...)` installed through
`model.model.encoder.set_input_embeddings` — what /root/reference/evaluation.py:72-86 builds.

Run here only (needs /root/reference):   python tools/make_goldens.py
The fixtures are data (inputs are regenerated from seeds by neuspeech1_amd.weights; outputs are
stored); no reference source travels.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuspeech1_amd.weights import LV2W, TINY, WHISPER_BASE, WHISPER_LARGE_V2, WhisperDims, make_lora_state, make_state_dict, synth_batch  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def ref_projection_module():
    spec = importlib.util.spec_from_file_location("ref_model_utils", os.path.join(REF, "utils", "model_utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.projection_module


def build_hf(dims: WhisperDims, sd_np):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    cfg = WhisperConfig(vocab_size=dims.vocab, num_mel_bins=80, d_model=dims.d, encoder_layers=dims.enc_layers,
                        decoder_layers=dims.dec_layers, encoder_attention_heads=dims.heads,
                        decoder_attention_heads=dims.heads, encoder_ffn_dim=dims.ffn, decoder_ffn_dim=dims.ffn,
                        max_source_positions=dims.src_pos, max_target_positions=dims.tgt_pos,
                        pad_token_id=dims.pad_id, bos_token_id=dims.bos_id, eos_token_id=dims.eos_id,
                        decoder_start_token_id=dims.start_id, attn_implementation="eager",
                        suppress_tokens=[], begin_suppress_tokens=[])
    model = WhisperForConditionalGeneration(cfg)
    conv1 = ref_projection_module()(config_name="base", meg_ch=dims.ch, d_model=model.model.encoder.conv2.in_channels)
    model.model.encoder.set_input_embeddings(conv1)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("proj_out" in m for m in missing), missing
    model.eval()
    return model


TRAINABLE = ("model.encoder.conv1.0.weight", "model.encoder.conv1.0.bias", "model.encoder.conv1.2.weight",
             "model.encoder.conv1.2.bias", "model.encoder.conv2.weight", "model.encoder.conv2.bias")


def train_golden(dims, tag, B, seed_w=42, seed_d=1234, full=True):
    sd_np = make_state_dict(dims, seed_w)
    model = build_hf(dims, sd_np)
    x, labels = synth_batch(dims, B, seed_d)
    for p in model.parameters():
        p.requires_grad_(False)
    named = dict(model.named_parameters())
    for k in TRAINABLE:
        named[k].requires_grad_(True)
    out = model(input_features=torch.from_numpy(x), labels=torch.from_numpy(labels))
    out.loss.backward()
    enc = out.encoder_last_hidden_state.detach().numpy()
    logits = out.logits.detach().numpy()
    g = {"loss": np.float32(out.loss.item()), "B": B, "seed_w": seed_w, "seed_d": seed_d, "labels": labels}
    top2 = np.sort(logits, -1)[..., -2:]
    g["top1_id"] = logits.argmax(-1).astype(np.int32)
    g["top_margin"] = (top2[..., 1] - top2[..., 0]).astype(np.float32)
    if full:
        g["enc"] = enc.astype(np.float32)
        g["logits"] = logits.astype(np.float32)
        for k in TRAINABLE:
            gr = named[k].grad.numpy().astype(np.float32)
            if gr.size > 100_000:      # keep fixtures small: leading block + Frobenius norm
                g["gradblock." + k] = gr[:48, :48].copy()
                g["gradnorm." + k] = np.float64(np.sqrt((gr.astype(np.float64) ** 2).sum()))
            else:
                g["grad." + k] = gr
    else:
        g["enc_slice"] = enc[:, ::97, :16].astype(np.float32)
        g["logits_slice"] = logits[:, :, :16].astype(np.float32)
        g["enc_sum"] = np.float64(enc.astype(np.float64).sum())
        g["enc_l2"] = np.float64(np.sqrt((enc.astype(np.float64) ** 2).sum()))
        for k in TRAINABLE:
            gr = named[k].grad.numpy()
            g["gradnorm." + k] = np.float64(np.sqrt((gr.astype(np.float64) ** 2).sum()))
            g["gradslice." + k] = gr.reshape(gr.shape[0], -1)[:8, :8].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"train_{tag}.npz"), **g)
    print(f"train_{tag}: loss {g['loss']:.6f}")
    return sd_np, x, labels


def lora_golden(dims, tag, B, r=32, alpha=64.0, decoder=False):
    """LoRA is pinned through merged-weight equivalence: the reference object run on W + (alpha/r) B A.
    decoder=True: adapters on every decoder projection as well (finetune.py --ft_full); logits kept as a slice."""
    sys.path.insert(0, os.path.join(ROOT))
    from oracle.whisper_meg_oracle import lora_merge
    sd_np = make_state_dict(dims, 42)
    lora_np = make_lora_state(dims, r, decoder=decoder)
    merged = lora_merge(sd_np, lora_np, alpha / r)
    model = build_hf(dims, merged)
    x, labels = synth_batch(dims, B, 1234)
    with torch.no_grad():
        out = model(input_features=torch.from_numpy(x), labels=torch.from_numpy(labels))
    lg = out.logits.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"lora_merged_{tag}.npz"), loss=np.float32(out.loss.item()),
                        logits=lg[:, :, ::7] if decoder else lg, r=r, alpha=alpha, B=B)
    print(f"lora_merged_{tag}: loss {out.loss.item():.6f}")


def adalora_golden(dims, tag, B, r=12, alpha=32.0):
    """AdaLoRA -- the reference's DEFAULT adapter (finetune.py:43,205-208: init_r=12, lora_alpha=32) -- pinned like LoRA:
    the reference object run on the merged weights W + alpha/(r+1e-5) * B (A * E) (peft SVDLinear.get_delta_weight),
    with a non-zero E so that the diag(E) path is live."""
    from oracle.whisper_meg_oracle import lora_merge
    sd_np = make_state_dict(dims, 42)
    lora_np = make_lora_state(dims, r, adalora=True, b_std=0.3)
    merged = lora_merge(sd_np, lora_np, alpha / (r + 1e-5))
    model = build_hf(dims, merged)
    x, labels = synth_batch(dims, B, 1234)
    with torch.no_grad():
        out = model(input_features=torch.from_numpy(x), labels=torch.from_numpy(labels))
        base = build_hf(dims, sd_np)(input_features=torch.from_numpy(x), labels=torch.from_numpy(labels))
    lg = out.logits.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"adalora_merged_{tag}.npz"), loss=np.float32(out.loss.item()),
                        loss_base=np.float32(base.loss.item()), logits=lg[:, :, ::3], r=r, alpha=alpha, B=B, b_std=0.3)
    print(f"adalora_merged_{tag}: loss {out.loss.item():.6f} (frozen model {base.loss.item():.6f})")


def lora_oracle_golden(dims, tag, B, r=32, alpha=64.0, seed_d=31):
    """Adapter gradients at a size where running the oracle inside a GPU test would take minutes (whisper-large-v2 at
    full depth): the ORACLE's loss and per-tensor gradient norms + leading blocks with LoRA r = 32 on every encoder
    projection (the oracle's adapter arithmetic is pinned on the reference object through merged weights by
    lora_merged_*.npz; the frozen path of the same config is pinned directly by train_<tag>.npz)."""
    from oracle import whisper_meg_oracle as O
    sd_np = make_state_dict(dims, 42)
    lora_np = make_lora_state(dims, r)
    x, labels = synth_batch(dims, B, seed_d)
    loss, logits, enc, grads = O.loss_and_grads(sd_np, lora_np, x, labels, dims, alpha / r)
    g = {"loss": np.float32(loss.item()), "B": B, "r": r, "alpha": alpha, "seed_d": seed_d, "labels": labels}
    names = sorted(grads)
    g["names"] = np.array(names)
    g["gradnorm"] = np.array([float(grads[k].double().norm()) for k in names], dtype=np.float64)
    g["gradblock"] = np.stack([np.pad(grads[k].reshape(grads[k].shape[0], -1)[:8, :8].numpy().astype(np.float32),
                                      ((0, 8 - min(8, grads[k].shape[0])), (0, 8 - min(8, grads[k].reshape(grads[k].shape[0], -1).shape[1]))))
                               for k in names])
    np.savez_compressed(os.path.join(OUT, f"lora_oracle_{tag}.npz"), **g)
    print(f"lora_oracle_{tag}: loss {g['loss']:.6f}, {len(names)} gradient tensors")


class _ForceTokens:
    """HF <= 4.37 ForceTokensLogitsProcessor [MEM: removed from the installed transformers 5.15]: at generation index
    idx = input_ids.shape[-1], if the map holds a non-None token for idx every score becomes -inf except that token's,
    which becomes 0.  The reference reaches it through generation_config.forced_decoder_ids
    (utils/load_model.py:1210-1256 -> super().generate)."""

    def __init__(self, forced):
        self.m = {int(k): v for k, v in forced}

    def __call__(self, input_ids, scores):
        tok = self.m.get(int(input_ids.shape[-1]))
        if tok is not None:
            scores = torch.full_like(scores, -float("inf"))
            scores[:, tok] = 0
        return scores


def decode_forced_golden(dims, tag, B, new_tokens):
    """forced_decoder_ids + non-empty suppress / begin-suppress lists through the reference object.  Era semantics
    (HF 4.31-4.35 generation/utils.py _get_logits_processor [MEM]): processors in the order repetition penalty,
    no-repeat-ngram, suppress_tokens, begin-suppress at begin_index = prompt_len + forced_decoder_ids[-1][0], force-tokens."""
    import transformers
    from transformers import LogitsProcessorList
    from transformers.generation.logits_process import SuppressTokensAtBeginLogitsProcessor
    sd_np = make_state_dict(dims, 42)
    model = build_hf(dims, sd_np)
    x, labels = synth_batch(dims, B, 1234)
    feats = torch.from_numpy(x)
    V = dims.vocab
    forced = [[1, None], [2, V - 6], [3, V - 5]]
    g0 = np.load(os.path.join(OUT, "decode_tiny.npz"))["greedy_rp"]
    suppress = sorted({int(g0[0, 5]), int(g0[1, 6]), 630, 34, V - 4, V - 3})
    begin_suppress = sorted({int(g0[0, 4]), int(g0[1, 4]), int(g0[2, 4]), 250, dims.eos_id})
    g = {"B": B, "new_tokens": new_tokens, "forced_idx": np.array([f[0] for f in forced]),
         "forced_tok": np.array([-1 if f[1] is None else f[1] for f in forced]), "suppress": np.array(suppress),
         "begin_suppress": np.array(begin_suppress)}
    gen = transformers.GenerationMixin.generate
    for pname, prompt in (("p1", torch.full((B, 1), dims.start_id, dtype=torch.long)), ("p4", torch.from_numpy(labels[:, :4].copy()))):
        P = prompt.shape[1]
        common = dict(do_sample=False, max_new_tokens=new_tokens, decoder_input_ids=prompt, suppress_tokens=suppress,
                      begin_suppress_tokens=None, pad_token_id=dims.pad_id, eos_token_id=dims.eos_id,
                      return_dict_in_generate=True, output_scores=True)
        mk = lambda: LogitsProcessorList([SuppressTokensAtBeginLogitsProcessor(begin_suppress, P + forced[-1][0]),  # noqa: E731
                                          _ForceTokens(forced)])
        g[f"{pname}.prompt"] = prompt.numpy()
        with torch.no_grad():
            g[f"{pname}.greedy"] = gen(model, feats, num_beams=1, logits_processor=mk(), **common).sequences.numpy()
            g[f"{pname}.greedy_rp"] = gen(model, feats, num_beams=1, repetition_penalty=5.0, no_repeat_ngram_size=2,
                                          logits_processor=mk(), **common).sequences.numpy()
            o = gen(model, feats, num_beams=5, repetition_penalty=5.0, no_repeat_ngram_size=2, logits_processor=mk(), **common)
            g[f"{pname}.beam5_rp"] = o.sequences.numpy()
            g[f"{pname}.beam5_rp_scores"] = o.sequences_scores.numpy().astype(np.float32)
            o = gen(model, feats, num_beams=5, logits_processor=mk(), **common)
            g[f"{pname}.beam5"] = o.sequences.numpy()
    np.savez_compressed(os.path.join(OUT, f"decode_{tag}_forced.npz"), **g)
    print(f"decode_{tag}_forced: p1 greedy head {g['p1.greedy'][:, :6].tolist()}")


def hf_checkpoint_golden(dims, tag, B):
    """a checkpoint WRITTEN BY STOCK HF save_pretrained (config.json + generation_config.json +
    model.safetensors; the stock 80-mel conv1 inside) -> load it back with stock HF, install the reference's
    projection_module (evaluation.py:72-86) and record logits + generated ids.  Only the outputs are committed; the test
    re-creates the directory with stock transformers (tests/hf_ckpt.py) and loads it with the build's from_pretrained."""
    import tempfile
    import transformers
    from transformers import WhisperForConditionalGeneration
    from tests.hf_ckpt import GEN_CFG, front_end_state, write_stock_hf_checkpoint
    with tempfile.TemporaryDirectory() as td:
        write_stock_hf_checkpoint(dims, td)
        model = WhisperForConditionalGeneration.from_pretrained(td, attn_implementation="eager")
        conv1 = ref_projection_module()(config_name="base", meg_ch=dims.ch, d_model=model.model.encoder.conv2.in_channels)
        conv1.load_state_dict({k: torch.from_numpy(v) for k, v in front_end_state(dims).items()})
        model.model.encoder.set_input_embeddings(conv1)
        model.eval()
        x, labels = synth_batch(dims, B, 1234)
        feats = torch.from_numpy(x)
        gc = model.generation_config
        assert list(gc.suppress_tokens) == GEN_CFG(dims)["suppress_tokens"], "generation_config.json did not round-trip"
        g = {"B": B}
        with torch.no_grad():
            out = model(input_features=feats, labels=torch.from_numpy(labels))
            g["loss"] = np.float32(out.loss.item())
            g["logits"] = out.logits.numpy().astype(np.float32)
            gen = transformers.GenerationMixin.generate
            prompt = torch.from_numpy(labels[:, :4].copy())
            # the lists come from the checkpoint's generation_config.json; max_length too (prompt 4 -> 4 + new tokens)
            common = dict(do_sample=False, decoder_input_ids=prompt, suppress_tokens=list(gc.suppress_tokens),
                          begin_suppress_tokens=list(gc.begin_suppress_tokens), max_length=gc.max_length,
                          pad_token_id=dims.pad_id, eos_token_id=dims.eos_id, return_dict_in_generate=True)
            g["greedy_rp"] = gen(model, feats, num_beams=1, repetition_penalty=5.0, no_repeat_ngram_size=2, **common).sequences.numpy()
            g["beam5_rp"] = gen(model, feats, num_beams=5, repetition_penalty=5.0, no_repeat_ngram_size=2, **common).sequences.numpy()
    np.savez_compressed(os.path.join(OUT, f"hf_ckpt_{tag}.npz"), **g)
    print(f"hf_ckpt_{tag}: loss {g['loss']:.6f}, greedy_rp width {g['greedy_rp'].shape[1]}")


def decode_golden(dims, tag, B, new_tokens, eos_variants=(34, 630), sequence_bias=None):
    import transformers
    sd_np = make_state_dict(dims, 42)
    model = build_hf(dims, sd_np)
    x, labels = synth_batch(dims, B, 1234)
    prompt = torch.from_numpy(labels[:, :4].copy())
    feats = torch.from_numpy(x)
    g = {"B": B, "new_tokens": new_tokens, "prompt": prompt.numpy()}
    common = dict(do_sample=False, max_new_tokens=new_tokens, decoder_input_ids=prompt, suppress_tokens=None,
                  begin_suppress_tokens=None, pad_token_id=dims.pad_id, eos_token_id=dims.eos_id,
                  return_dict_in_generate=True, output_scores=True)
    gen = transformers.GenerationMixin.generate
    with torch.no_grad():
        def margins(o):
            """per generated position: processed top-1 minus top-2 score of the reference's own decision (fp32).  A greedy
            decision whose margin is below fp16 resolution cannot be required of an fp16 path (tests: ids must match up
            to the first position whose margin is < 0.03; everything after such a flip is a different sequence)."""
            t2 = [torch.topk(sc, 2, dim=-1).values for sc in o.scores]
            return torch.stack([(v[:, 0] - v[:, 1]) for v in t2], 1).numpy().astype(np.float32)

        o = gen(model, feats, num_beams=1, **common)
        g["greedy"] = o.sequences.numpy()
        g["greedy_margin"] = margins(o)
        g["greedy_step0_scores"] = o.scores[0].numpy()[:, :64].astype(np.float32)
        o = gen(model, feats, num_beams=1, repetition_penalty=5.0, no_repeat_ngram_size=2, **common)
        g["greedy_rp"] = o.sequences.numpy()
        g["greedy_rp_margin"] = margins(o)
        o = gen(model, feats, num_beams=5, repetition_penalty=5.0, no_repeat_ngram_size=2, **common)
        g["beam5_rp"] = o.sequences.numpy()
        g["beam5_rp_scores"] = o.sequences_scores.numpy().astype(np.float32)
        o = gen(model, feats, num_beams=5, **common)
        g["beam5"] = o.sequences.numpy()
        g["beam5_scores"] = o.sequences_scores.numpy().astype(np.float32)
        if sequence_bias is not None:
            # HF SequenceBiasLogitsProcessor through the reference object: single tokens, 2- and 3-token sequences
            g["sequence_bias_keys"] = np.array([",".join(map(str, k)) for k in sequence_bias])
            g["sequence_bias_vals"] = np.array(list(sequence_bias.values()), dtype=np.float64)
            o = gen(model, feats, num_beams=1, sequence_bias=dict(sequence_bias), **common)
            g["greedy_sb"] = o.sequences.numpy()
            o = gen(model, feats, num_beams=1, repetition_penalty=5.0, no_repeat_ngram_size=2, sequence_bias=dict(sequence_bias), **common)
            g["greedy_rp_sb"] = o.sequences.numpy()
            o = gen(model, feats, num_beams=5, repetition_penalty=5.0, no_repeat_ngram_size=2, sequence_bias=dict(sequence_bias), **common)
            g["beam5_rp_sb"] = o.sequences.numpy()
            g["beam5_rp_sb_scores"] = o.sequences_scores.numpy().astype(np.float32)
            o = gen(model, feats, num_beams=5, sequence_bias=dict(sequence_bias), **common)
            g["beam5_sb"] = o.sequences.numpy()
            g["beam5_sb_scores"] = o.sequences_scores.numpy().astype(np.float32)
        # EOS variants: declare a frequently generated token id as EOS so rows / hypotheses really finish
        for eos in eos_variants:
            c2 = dict(common, eos_token_id=eos)
            o = gen(model, feats, num_beams=1, **c2)
            g[f"greedy_eos{eos}"] = o.sequences.numpy()
            o = gen(model, feats, num_beams=5, **c2)
            g[f"beam5_eos{eos}"] = o.sequences.numpy()
            g[f"beam5_eos{eos}_scores"] = o.sequences_scores.numpy().astype(np.float32)
            o = gen(model, feats, num_beams=5, repetition_penalty=5.0, no_repeat_ngram_size=2, **c2)
            g[f"beam5_rp_eos{eos}"] = o.sequences.numpy()
            g[f"beam5_rp_eos{eos}_scores"] = o.sequences_scores.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"decode_{tag}.npz"), **g)
    print(f"decode_{tag}: greedy tail {g['greedy'][:, -4:].tolist()}")


def decode_hyps_golden(dims, tag, B, new_tokens, eos_variants=(), src=None, sequence_bias=None, forced=False):
    """the reference object's top-`num_beams` FINISHED hypotheses (ids + length-normalised scores) of every
    beam-search golden, so that a beam row which leaves hypothesis 0 at an fp16 near-tie can be required to BE one of the
    reference's own alternatives (evaluation.py:369-386 -> GenerationMixin.generate with num_return_sequences = num_beams).
    Hypothesis 0 must reproduce the ids already stored in decode_<tag>.npz (asserted here).  -> decode_<tag>_hyps.npz"""
    import transformers
    sd_np = make_state_dict(dims, 42)
    model = build_hf(dims, sd_np)
    x, labels = synth_batch(dims, B, 1234)
    feats = torch.from_numpy(x)
    old = np.load(os.path.join(OUT, f"decode_{src or tag}.npz"))
    gen = transformers.GenerationMixin.generate
    nb = 5
    g = {"B": B, "new_tokens": new_tokens, "num_beams": nb}

    def run(name, prompt, **kw):
        common = dict(do_sample=False, max_new_tokens=new_tokens, decoder_input_ids=prompt, suppress_tokens=None,
                      begin_suppress_tokens=None, pad_token_id=dims.pad_id, eos_token_id=dims.eos_id,
                      return_dict_in_generate=True, output_scores=True, num_beams=nb, num_return_sequences=nb)
        common.update(kw)
        with torch.no_grad():
            o = gen(model, feats, **common)
        seqs = o.sequences.numpy().reshape(B, nb, -1)
        sc = o.sequences_scores.numpy().astype(np.float32).reshape(B, nb)
        ref = old[name]
        Lm = min(ref.shape[1], seqs.shape[2])
        assert np.array_equal(seqs[:, 0, :Lm], ref[:, :Lm]), (name, seqs[:, 0].tolist(), ref.tolist())
        g[name + "_hyps"], g[name + "_hyp_scores"] = seqs, sc
        print(f"decode_{tag}_hyps {name}: scores row 0 {sc[0].round(4).tolist()}")

    rp = dict(repetition_penalty=5.0, no_repeat_ngram_size=2)
    if forced:
        from transformers import LogitsProcessorList
        from transformers.generation.logits_process import SuppressTokensAtBeginLogitsProcessor
        fz = [[int(i), None if int(t) < 0 else int(t)] for i, t in zip(old["forced_idx"], old["forced_tok"])]
        for pname in ("p1", "p4"):
            prompt = torch.from_numpy(old[pname + ".prompt"])
            P = prompt.shape[1]
            mk = lambda: LogitsProcessorList([SuppressTokensAtBeginLogitsProcessor(old["begin_suppress"].tolist(), P + fz[-1][0]),  # noqa: E731
                                              _ForceTokens(fz)])
            run(f"{pname}.beam5_rp", prompt, suppress_tokens=old["suppress"].tolist(), logits_processor=mk(), **rp)
            run(f"{pname}.beam5", prompt, suppress_tokens=old["suppress"].tolist(), logits_processor=mk())
    else:
        prompt = torch.from_numpy(labels[:, :4].copy())
        sfx = ""
        extra = {}
        if sequence_bias is not None:
            sfx, extra = "_sb", dict(sequence_bias=dict(sequence_bias))
        run("beam5_rp" + sfx, prompt, **rp, **extra)
        run("beam5" + sfx, prompt, **extra)
        for eos in eos_variants:
            run(f"beam5_eos{eos}", prompt, eos_token_id=eos)
            run(f"beam5_rp_eos{eos}", prompt, eos_token_id=eos, **rp)
    np.savez_compressed(os.path.join(OUT, f"decode_{tag}_hyps.npz"), **g)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    what = sys.argv[1:] or ["train", "lora", "decode"]
    if "train" in what:
        train_golden(TINY, "tiny", B=2, full=True)
        train_golden(WHISPER_BASE, "base208", B=2, full=False)
        train_golden(WhisperDims(ch=273), "base273", B=1, full=False)
    if "lora" in what:
        lora_golden(TINY, "tiny", B=2)
    if "lora_full" in what:
        lora_golden(TINY, "tiny_full", B=2, decoder=True)
    if "decode" in what:
        decode_golden(TINY, "tiny", B=3, new_tokens=24)
    if "decode_sb" in what:
        # biases chosen on the unbiased greedy output of decode_tiny.npz: penalise its most frequent tokens, reward a few
        # others, and bias continuations of 2- / 3-token contexts that really occur
        g0 = np.load(os.path.join(OUT, "decode_tiny.npz"))["greedy_rp"]
        sb = {(630,): -4.0, (34,): -2.5, (17,): 3.0, (250,): 2.0}
        for b in range(g0.shape[0]):
            sb[(int(g0[b, 6]), int(g0[b, 7]), 99 + b)] = 9.0          # after (t6, t7) strongly prefer token 99 + b
            sb[(int(g0[b, 9]), int(g0[b, 10]))] = -8.0                # and break the original continuation t9 -> t10
            sb[(int(g0[b, 5]), int(g0[b, 6]))] = -1.5
        decode_golden(TINY, "tiny_sb", B=3, new_tokens=24, eos_variants=(), sequence_bias=sb)
    if "decode_base" in what:
        # BASELINE dims (whisper-base, 208-ch): token-id parity at the size the metric is quoted on
        decode_golden(WHISPER_BASE, "base208", B=2, new_tokens=16, eos_variants=())
    if "decode_base273" in what:
        # BASELINE configs[3] (README.md:60-64: Schoffelen, --eeg_ch=273): beam-5 + penalties at whisper-base, 273 channels
        decode_golden(WhisperDims(ch=273), "base273", B=2, new_tokens=16, eos_variants=())
    if "adalora" in what:
        adalora_golden(TINY, "tiny", B=2)
    if "lv2w" in what:
        # BASELINE configs[4] WIDTH (whisper-large-v2: d=1280, 20 heads, ffn 5120, 273 channels) at 2+2 layers
        train_golden(LV2W, "lv2w", B=1, full=False)
        decode_golden(LV2W, "lv2w", B=2, new_tokens=12, eos_variants=())
    if "lv2" in what:
        # BASELINE configs[4] at FULL depth (whisper-large-v2: 32 + 32 layers, d 1280, 273 channels), B = 1: the frozen path on
        # the reference object, the adapter gradients on the oracle
        train_golden(WHISPER_LARGE_V2, "lv2", B=1, full=False)
        lora_oracle_golden(WHISPER_LARGE_V2, "lv2", B=1)
    if "forced" in what:
        decode_forced_golden(TINY, "tiny", B=3, new_tokens=20)
    if "hyps" in what:
        # top-num_beams finished hypotheses of every beam golden (decode_*_hyps.npz)
        decode_hyps_golden(TINY, "tiny", B=3, new_tokens=24, eos_variants=(34, 630))
        sbg = np.load(os.path.join(OUT, "decode_tiny_sb.npz"))
        sb = {tuple(int(t) for t in str(k).split(",")): float(v) for k, v in zip(sbg["sequence_bias_keys"], sbg["sequence_bias_vals"])}
        decode_hyps_golden(TINY, "tiny_sb", B=3, new_tokens=24, sequence_bias=sb)
        decode_hyps_golden(TINY, "tiny_forced", B=3, new_tokens=20, forced=True)
        decode_hyps_golden(WHISPER_BASE, "base208", B=2, new_tokens=16)
        decode_hyps_golden(WhisperDims(ch=273), "base273", B=2, new_tokens=16)
        decode_hyps_golden(LV2W, "lv2w", B=2, new_tokens=12)
    if "hf_ckpt" in what:
        hf_checkpoint_golden(TINY, "tiny", B=2)


def reader_golden():
    """G6: run the REFERENCE reader / collator / matchers (imported from /root/reference with the four absent
    third-party modules stubbed; none is touched on the EEG path) on seeded synthetic files."""
    import json
    import tempfile
    import types
    for name in ("jsonlines", "librosa", "soundfile", "zhconv", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["zhconv"].convert = lambda t, _: t

    class _JL:
        def __init__(self, path, mode="r"):
            self.f = open(path, mode)
        def __enter__(self):
            return self
        def __exit__(self, *a):
            self.f.close()
        def __iter__(self):
            return (json.loads(l) for l in self.f if l.strip())
        def write(self, obj):
            self.f.write(json.dumps(obj) + "\n")
    sys.modules["jsonlines"].open = lambda p, mode="r": _JL(p, mode)
    sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[m]
    import importlib
    ref_reader = importlib.import_module("utils.reader")
    ref_data = importlib.import_module("utils.data_utils")
    from neuspeech1_amd.synthetic import SyntheticProcessor
    from neuspeech1_amd.weights import WHISPER_BASE
    proc = SyntheticProcessor(WHISPER_BASE)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        cases = [("gwilliams", 224, 208), ("schoffelen", 301, 273), ("other", 100, 208)]
        for name, ch_file, modal_ch in cases:
            os.makedirs(os.path.join(td, name), exist_ok=True)
            rows = []
            for n in (700, 6000, 7321):
                rng = np.random.default_rng(ch_file * 100000 + n)
                x = rng.standard_normal((ch_file, n))
                p = os.path.join(td, name, f"s{n}.npy")
                np.save(p, x)
                rows.append({"eeg": {"path": p}, "sentence": f"hello {name} {n}", "language": "English", "duration": n / 200})
            jl = os.path.join(td, f"{name}.jsonl")
            with open(jl, "w") as f:
                for r in rows:
                    f.write(json.dumps(r) + "\n")
            ds = ref_reader.CustomDataset(data_list_path=jl, processor=proc, modal="eeg", modal_ch=modal_ch, mode="val",
                                          sample_rate=200, orig_sample_rate=200, language="English", timestamps=False,
                                          min_duration=0.5, max_duration=30)
            items = [ds[i] for i in range(len(ds))]
            for i, it in enumerate(items):
                a = it["input_features"][0]
                nz = np.nonzero(np.abs(a).sum(0))[0]
                out[f"{name}.{i}.shape"] = np.array(a.shape)
                out[f"{name}.{i}.sum"] = np.float64(a.sum())
                out[f"{name}.{i}.abs"] = np.float64(np.abs(a).sum())
                out[f"{name}.{i}.last_nz"] = np.int64(nz[-1])
                out[f"{name}.{i}.rows_nz"] = np.int64((np.abs(a).sum(1) > 0).sum())
                out[f"{name}.{i}.labels"] = np.array(it["labels"])
            batch = ref_data.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)(items)
            out[f"{name}.batch_dtype"] = np.array(str(batch["input_features"].dtype))
            out[f"{name}.batch_sum"] = np.float64(batch["input_features"].double().sum().item())
            out[f"{name}.batch_labels"] = batch["labels"].numpy()
        # timestamp labels (finetune.py's default --timestamps=True; reference reader :347-400): sentence and word level
        rng = np.random.default_rng(99)
        rows = []
        for k in range(4):
            n = 1200 + 300 * k
            p = os.path.join(td, "gwilliams", f"ts{k}.npy")
            np.save(p, rng.standard_normal((224, n)))
            sents, t0 = [], 0.0
            for si in range(1 + k % 3):
                words, t = [], t0 + float(rng.integers(0, 50)) / 100
                for wi in range(2 + si):
                    dur = float(rng.integers(5, 60)) / 100           # odd and even centiseconds both occur
                    words.append({"start": round(t, 2), "end": round(t + dur, 2), "word": f"w{k}{si}{wi}"})
                    t += dur + float(rng.integers(0, 9)) / 100
                sents.append({"start": words[0]["start"], "end": words[-1]["end"], "text": " ".join(w["word"] for w in words),
                              "words": words})
                t0 = t
            rows.append({"eeg": {"path": p}, "sentence": " ".join(s_["text"] for s_ in sents), "sentences": sents,
                         "language": "English", "duration": n / 200})
        jl = os.path.join(td, "ts.jsonl")
        with open(jl, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
        out["ts.rows"] = np.array(json.dumps(rows))
        for level in ("sentences", "words"):
            ds = ref_reader.CustomDataset(data_list_path=jl, processor=proc, modal="eeg", modal_ch=208, mode="val", level=level,
                                          sample_rate=200, orig_sample_rate=200, language="English", timestamps=True,
                                          min_duration=0.5, max_duration=30)
            items = [ds[i] for i in range(len(ds))]
            for i, it in enumerate(items):
                out[f"ts.{level}.{i}.labels"] = np.array(it["labels"])
                out[f"ts.{level}.{i}.sum"] = np.float64(it["input_features"][0].sum())
            batch = ref_data.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)(items)
            out[f"ts.{level}.batch_labels"] = batch["labels"].numpy()
    np.savez_compressed(os.path.join(OUT, "reader.npz"), **out)
    print("reader golden:", len(out), "entries")
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[m]
    sys.path.remove(REF)


if __name__ == "__main__" and "reader" in (sys.argv[1:] or ["reader"]):
    reader_golden()
