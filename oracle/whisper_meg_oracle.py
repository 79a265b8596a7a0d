"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (neuspeech1_amd/) never does and fails loudly when
its HIP library is missing.

A plain fp32 torch-CPU restatement of NeuSpeech's Whisper MEG->text hot path,
written from the formulas (not copied): every function cites the reference
file:line (relative to the reference tree) or the HuggingFace file (HF: =
transformers/models/whisper/modeling_whisper.py, transformers/generation/*)
whose arithmetic it follows.

Pinning: the reference ships no tests or golden vectors, and its
arithmetic lives in third-party `transformers` (unpinned; 5.15.0 installed) and
`peft` (absent).  This oracle is therefore pinned against OUTPUTS OF THE
REFERENCE OBJECT ITSELF generated in the build container: stock
`transformers.WhisperForConditionalGeneration` (eager attention) with the
reference's own `utils/model_utils.projection_module` installed via
`set_input_embeddings` — exactly what evaluation.py:72-86 builds — by
tools/make_goldens.py; the vectors are committed under tests/golden/ and checked
by tests/test_oracle_golden.py.  LoRA (peft) semantics cannot be imported and
are pinned only by merged-weight equivalence (see lora_merge below).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def T(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))


# --------------------------------------------------------------------------- front-end
def frontend(sd, x):
    """utils/model_utils.py:11-16 (Conv1d k3 p1 -> GELU -> Conv1d k3 s2 p1) then
    utils/load_model.py:410-416: gelu(conv1(x)), gelu(conv2(.)), permute, + embed_positions."""
    e = "model.encoder."
    if e + "conv1.weight" in sd:                     # projection_module('replace'): one stride-2 conv (model_utils.py:19-21)
        h = F.conv1d(x, sd[e + "conv1.weight"], sd[e + "conv1.bias"], stride=2, padding=1)
    else:
        h = F.conv1d(x, sd[e + "conv1.0.weight"], sd[e + "conv1.0.bias"], stride=1, padding=1)
        h = F.gelu(h)
        h = F.conv1d(h, sd[e + "conv1.2.weight"], sd[e + "conv1.2.bias"], stride=2, padding=1)
    h = F.gelu(h)                                   # the encoder's own gelu(conv1(x))
    h = F.gelu(F.conv1d(h, sd[e + "conv2.weight"], sd[e + "conv2.bias"], stride=2, padding=1))
    h = h.permute(0, 2, 1)
    return h + sd[e + "embed_positions.weight"]


def _lora_lin(x, sd, lora, name, scale):
    """y = W x + b (+ scale * B(A x))   peft lora.Linear.forward at dropout 0 (finetune.py:210-212)."""
    y = F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))
    if lora is not None and (name + ".lora_A.weight") in lora:
        u = F.linear(x, lora[name + ".lora_A.weight"])
        if (name + ".lora_E.weight") in lora:      # AdaLoRA (peft SVDLinear): B((A x) * E) * alpha / (r + 1e-5)
            u = u * lora[name + ".lora_E.weight"].reshape(-1)
        y = y + scale * F.linear(u, lora[name + ".lora_B.weight"])
    return y


def adalora_orth_reg(lora):
    """peft AdaLoraModel.forward: mean over every lora_A / lora_B of || P P^T - I ||_F (A) resp. || P^T P - I ||_F (B);
    the caller multiplies by orth_reg_weight (finetune.py:207: 0.5)."""
    tot, num = 0.0, 0
    for k, p in lora.items():
        if k.endswith("lora_A.weight") or k.endswith("lora_B.weight"):
            cov = p @ p.T if "lora_A" in k else p.T @ p
            tot = tot + torch.norm(cov - torch.eye(cov.shape[0]), p="fro")
            num += 1
    return tot / num


def attention(sd, lora, prefix, x, kv, heads, mask, scale):
    """HF:241-357 WhisperAttention + :215-238 eager attention: q scaled by head_dim^-0.5 BEFORE the
    head split, k without bias, softmax(q k^T + mask) v, out_proj."""
    B, Lq, d = x.shape
    dh = d // heads
    q = _lora_lin(x, sd, lora, prefix + ".q_proj", scale) * dh ** -0.5
    k = _lora_lin(kv, sd, lora, prefix + ".k_proj", scale)
    v = _lora_lin(kv, sd, lora, prefix + ".v_proj", scale)
    q = q.view(B, Lq, heads, dh).transpose(1, 2)
    k = k.view(B, -1, heads, dh).transpose(1, 2)
    v = v.view(B, -1, heads, dh).transpose(1, 2)
    s = q @ k.transpose(2, 3)
    if mask is not None:
        s = s + mask
    p = s.softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, d)
    return _lora_lin(o, sd, lora, prefix + ".out_proj", scale)


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def encoder(sd, x, dims, lora=None, scale=0.0):
    """utils/load_model.py:410-468 + HF:360-413 (pre-LN encoder layer; the fp16 clamp never fires in fp32)."""
    h = frontend(sd, x)
    for i in range(dims.enc_layers):
        p = f"model.encoder.layers.{i}."
        h = h + attention(sd, lora, p + "self_attn", _ln(sd, p + "self_attn_layer_norm", h), _ln(sd, p + "self_attn_layer_norm", h),
                          dims.heads, None, scale)
        m = _ln(sd, p + "final_layer_norm", h)
        m = _lora_lin(F.gelu(_lora_lin(m, sd, lora, p + "fc1", scale)), sd, lora, p + "fc2", scale)
        h = h + m
    return _ln(sd, "model.encoder.layer_norm", h)


def causal_mask(Lq, Lk, dtype=torch.float32):
    """utils/load_model.py:101-115 (_make_causal_mask): finfo.min above the diagonal, offset by the cache length."""
    i = torch.arange(Lq)[:, None]
    j = torch.arange(Lk)[None, :]
    m = torch.zeros(Lq, Lk, dtype=dtype)
    m.masked_fill_(j > i + (Lk - Lq), torch.finfo(dtype).min)
    return m


def decoder(sd, dec_ids, enc, dims, pos0=0, lora=None, scale=0.0):
    """utils/load_model.py:645-749 + HF:416-506: embed_tokens + positions[pos0:pos0+L], then per layer
    causal self-attn, cross-attn over encoder states, GELU MLP (all pre-LN), final LayerNorm."""
    dd = "model.decoder."
    L = dec_ids.shape[1]
    h = sd[dd + "embed_tokens.weight"][dec_ids] + sd[dd + "embed_positions.weight"][pos0:pos0 + L]
    mask = causal_mask(L, L)
    for i in range(dims.dec_layers):
        p = f"{dd}layers.{i}."
        x = _ln(sd, p + "self_attn_layer_norm", h)
        # adapters on the decoder projections only with finetune.py --ft_full (:191-192); _lora_lin is the plain linear
        # for modules without lora_A / lora_B entries
        h = h + attention(sd, lora, p + "self_attn", x, x, dims.heads, mask, scale)
        h = h + attention(sd, lora, p + "encoder_attn", _ln(sd, p + "encoder_attn_layer_norm", h), enc, dims.heads, None, scale)
        m = _ln(sd, p + "final_layer_norm", h)
        h = h + _lora_lin(F.gelu(_lora_lin(m, sd, lora, p + "fc1", scale)), sd, lora, p + "fc2", scale)
    return _ln(sd, dd + "layer_norm", h)


def shift_tokens_right(labels, pad_id, start_id):
    """HF:68-81, called at utils/load_model.py:1025-1029."""
    out = torch.zeros_like(labels)
    out[:, 1:] = labels[:, :-1]
    out[:, 0] = start_id
    return out.masked_fill(out == -100, pad_id)


def forward(sd, x, dims, labels=None, dec_ids=None, lora=None, scale=0.0):
    """utils/load_model.py:976-1054: encoder, decoder, tied proj_out, CrossEntropyLoss (mean over labels != -100)."""
    if dec_ids is None:
        dec_ids = shift_tokens_right(labels, dims.pad_id, dims.start_id)
    enc = encoder(sd, x, dims, lora, scale)
    hid = decoder(sd, dec_ids, enc, dims, lora=lora, scale=scale)
    logits = F.linear(hid, sd["model.decoder.embed_tokens.weight"])
    loss = None
    if labels is not None:
        loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=-100)
    return loss, logits, enc


def to_torch(sd_np, requires_grad=()):
    out = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).clone()
        if any(k.startswith(p) or k.endswith(p) for p in requires_grad):
            t.requires_grad_(True)
        out[k] = t
    return out


TRAINABLE_CONV = ("model.encoder.conv1.0.weight", "model.encoder.conv1.0.bias", "model.encoder.conv1.2.weight",
                  "model.encoder.conv1.2.bias", "model.encoder.conv2.weight", "model.encoder.conv2.bias",
                  "model.encoder.conv1.weight", "model.encoder.conv1.bias")


def loss_and_grads(sd_np, lora_np, x_np, labels_np, dims, scale, orth_reg_weight=0.0):
    """Loss + gradients of the reference's trainable set: LoRA A/B on the encoder's q/k/v/out/fc1/fc2 and the
    three conv modules (finetune.py:187-212: target_modules + modules_to_save)."""
    sd = to_torch(sd_np, requires_grad=TRAINABLE_CONV)
    lora = to_torch(lora_np, requires_grad=("lora_A.weight", "lora_B.weight", "lora_E.weight")) if lora_np else None
    loss, logits, enc = forward(sd, T(x_np), dims, labels=T(labels_np), lora=lora, scale=scale)
    if orth_reg_weight and lora:
        loss = loss + orth_reg_weight * adalora_orth_reg(lora)
    loss.backward()
    grads = {k: sd[k].grad for k in TRAINABLE_CONV if k in sd}
    if lora:
        grads.update({k: v.grad for k, v in lora.items()})
    return loss.detach(), logits.detach(), enc.detach(), grads


def lora_merge(sd_np, lora_np, scale):
    """merge_and_unload (evaluation.py:88-89, merge_lora.py:43-44): W <- W + scale * B A  (LoRA), resp.
    W <- W + scale * B (A * E) with scale = alpha / (r + 1e-5)  (AdaLoRA, peft SVDLinear.get_delta_weight; the
    reference's default adapter, finetune.py:43,205-208)."""
    out = dict(sd_np)
    for k in lora_np:
        if k.endswith(".lora_A.weight"):
            base = k[: -len(".lora_A.weight")]
            a = lora_np[k]
            if (base + ".lora_E.weight") in lora_np:
                a = a * lora_np[base + ".lora_E.weight"].reshape(-1, 1)
            out[base + ".weight"] = (sd_np[base + ".weight"] + scale * lora_np[base + ".lora_B.weight"] @ a).astype(np.float32)
    return out


# --------------------------------------------------------------------------- optimizer
def adamw_reference(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.AdamW single-tensor update (optim='adamw_torch', finetune.py:247); step is 1-based."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = v.sqrt() / math.sqrt(1 - b2 ** step) + eps
    p = p - (lr / (1 - b1 ** step)) * m / denom
    return p, m, v


def linear_schedule(step, warmup, total):
    """HF get_linear_schedule_with_warmup lambda (Seq2SeqTrainingArguments defaults, finetune.py:236-237)."""
    if total <= 0:
        return 1.0
    if step < warmup:
        return step / max(1, warmup)
    return max(0.0, (total - step) / max(1, total - warmup))


# --------------------------------------------------------------------------- decode
def _decoder_logits_last(sd, ids, enc, dims):
    hid = decoder(sd, ids, enc, dims)
    return F.linear(hid[:, -1], sd["model.decoder.embed_tokens.weight"])


def repetition_penalty_(scores, ids, penalty):
    """HF:generation/logits_process.py:306-414: s<0 ? s*p : s/p on every token already in the sequence."""
    g = torch.gather(scores, 1, ids)
    g = torch.where(g < 0, g * penalty, g / penalty)
    scores.scatter_(1, ids, g)
    return scores


def no_repeat_ngram_(scores, ids, n):
    """HF:generation/logits_process.py:1073-1141: ban every token that would complete an already-seen n-gram."""
    Bn, cur = ids.shape
    if cur + 1 < n:
        return scores
    for b in range(Bn):
        seq = ids[b].tolist()
        prefix = tuple(seq[cur - (n - 1):]) if n > 1 else ()
        for s in range(cur - n + 1):
            if tuple(seq[s:s + n - 1]) == prefix:
                scores[b, seq[s + n - 1]] = -float("inf")
    return scores


def suppress_(scores, suppress_tokens, begin_suppress, cur_len, begin_index):
    """HF:generation/logits_process.py:1816-1906 (SuppressTokens, SuppressTokensAtBegin)."""
    if suppress_tokens:
        scores[:, suppress_tokens] = -float("inf")
    if begin_suppress and cur_len == begin_index:
        scores[:, begin_suppress] = -float("inf")
    return scores


def force_tokens_(scores, forced_decoder_ids, cur_len):
    """HF ForceTokensLogitsProcessor (generation/logits_process.py of the reference's era; LAST processor): at
    generation index cur_len, a non-None forced token gets score 0 and every other token -inf.  The reference hands
    generation_config.forced_decoder_ids to it through utils/load_model.py:1210-1256,1314-1322."""
    if forced_decoder_ids:
        tok = {int(i): t for i, t in forced_decoder_ids}.get(cur_len)
        if tok is not None:
            scores[:] = -float("inf")
            scores[:, tok] = 0.0
    return scores


def begin_index_for(prompt_len, forced_decoder_ids):
    """HF generation/utils.py _get_logits_processor of the reference's era: the begin-suppress list applies at
    prompt_len (+ forced_decoder_ids[-1][0] when forced ids are set)."""
    return prompt_len + (int(forced_decoder_ids[-1][0]) if forced_decoder_ids else 0)


def greedy(sd, x, dims, prompt, max_new_tokens, repetition_penalty=1.0, no_repeat_ngram_size=0,
           suppress_tokens=(), begin_suppress_tokens=(), eos_id=None, forced_decoder_ids=None):
    """HF:generation/utils.py:2783-2975 (greedy search): processors act on raw last-position logits
    (fp32), argmax, finished rows emit pad.  Returns prompt + new tokens (GenerationMixin semantics)."""
    eos_id = dims.eos_id if eos_id is None else eos_id
    enc = encoder(sd, x, dims)
    ids = prompt.clone()
    Bn = ids.shape[0]
    done = torch.zeros(Bn, dtype=torch.bool)
    begin = begin_index_for(prompt.shape[1], forced_decoder_ids)
    for _ in range(max_new_tokens):
        s = _decoder_logits_last(sd, ids, enc, dims).float().clone()
        if repetition_penalty != 1.0:
            repetition_penalty_(s, ids, repetition_penalty)
        if no_repeat_ngram_size > 0:
            no_repeat_ngram_(s, ids, no_repeat_ngram_size)
        suppress_(s, list(suppress_tokens), list(begin_suppress_tokens), ids.shape[1], begin)
        force_tokens_(s, forced_decoder_ids, ids.shape[1])
        nxt = s.argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, dims.pad_id), nxt)
        ids = torch.cat([ids, nxt[:, None]], 1)
        done = done | (nxt == eos_id)
        if bool(done.all()):
            break
    return ids


def beam_search(sd, x, dims, prompt, num_beams, max_new_tokens, repetition_penalty=1.0, no_repeat_ngram_size=0,
                suppress_tokens=(), begin_suppress_tokens=(), length_penalty=1.0, eos_id=None, forced_decoder_ids=None):
    """HF:generation/utils.py:3208-3545 (_beam_search, do_sample=False, early_stopping=False) with its helpers
    :3077-3129 (_get_top_k_continuations), :3131-3152 (_get_running_beams_for_next_iteration), :3154-3204
    (_update_finished_beams), :3008-3053 (_check_early_stop_heuristic), :3055-3075 (loop condition).

    Per step at length cur: log_softmax -> processors (on log-probs) -> + running beam scores -> top-(2*beams)
    over beams*V.  A candidate "hits" when its token is EOS or when cur+1 reaches max_length.  The next running
    beams are the best `beams` non-hit candidates.  Hit candidates of rank < beams enter the finished set with score
    / (cur+1-prompt)^lp -- unless the row's early-stop heuristic has already been satisfied (sticky) -- and the best
    `beams` finished hypotheses are kept.  Heuristic (after cur += 1): the row stays open while some finished slot is
    empty or running_best / (cur-prompt)^lp > worst finished score.  The loop ends when no row is open or every
    candidate hit (max length).  Returns finished hypothesis 0 of each row, padded with pad_id."""
    eos_id = dims.eos_id if eos_id is None else eos_id
    enc = encoder(sd, x, dims)
    Bn, P = prompt.shape
    V = dims.vocab
    nb = num_beams
    max_len = P + max_new_tokens
    enc_b = enc.repeat_interleave(nb, 0)
    running = torch.full((Bn, nb, max_len), dims.pad_id, dtype=torch.long)
    running[:, :, :P] = prompt[:, None, :]
    run_scores = torch.zeros(Bn, nb)
    run_scores[:, 1:] = -1e9
    fin_seqs = running.clone()
    fin_scores = torch.full((Bn, nb), -1e9)
    fin_done = torch.zeros(Bn, nb, dtype=torch.bool)
    open_row = torch.ones(Bn, 1, dtype=torch.bool)          # is_early_stop_heuristic_unsatisfied
    rank_ok = (torch.arange(2 * nb) < nb)[None, :]
    cur = P
    while True:
        flat = running[:, :, :cur].reshape(Bn * nb, cur)
        logits = _decoder_logits_last(sd, flat, enc_b, dims).float()
        lp = F.log_softmax(logits, -1)
        if repetition_penalty != 1.0:
            repetition_penalty_(lp, flat, repetition_penalty)
        if no_repeat_ngram_size > 0:
            no_repeat_ngram_(lp, flat, no_repeat_ngram_size)
        suppress_(lp, list(suppress_tokens), list(begin_suppress_tokens), cur, begin_index_for(P, forced_decoder_ids))
        force_tokens_(lp, forced_decoder_ids, cur)
        tot = (lp.view(Bn, nb, V) + run_scores[:, :, None]).view(Bn, nb * V)
        top, idx = torch.topk(tot, 2 * nb, dim=1)
        beam_idx = idx // V
        tok = idx % V
        cand = torch.gather(running, 1, beam_idx[:, :, None].expand(-1, -1, max_len)).clone()
        cand[:, :, cur] = tok
        hits = (tok == eos_id) | (cur + 1 >= max_len)
        # next running beams
        run_cand = top + hits.float() * -1.0e9
        rs, ri = torch.topk(run_cand, nb, dim=1)
        running = torch.gather(cand, 1, ri[:, :, None].expand(-1, -1, max_len))
        run_scores = rs
        # finished set
        just = hits & rank_ok
        fc = top / float(cur + 1 - P) ** length_penalty
        fc = fc + (~open_row).float() * -1.0e9
        fc = fc + (~just).float() * -1.0e9
        all_scores = torch.cat([fin_scores, fc], 1)
        all_seqs = torch.cat([fin_seqs, cand], 1)
        all_done = torch.cat([fin_done, just], 1)
        ts, ti = torch.topk(all_scores, nb, dim=1)
        fin_scores = ts
        fin_seqs = torch.gather(all_seqs, 1, ti[:, :, None].expand(-1, -1, max_len))
        fin_done = torch.gather(all_done, 1, ti)
        cur += 1
        best_run = run_scores[:, :1] / float(cur - P) ** length_penalty
        worst = torch.where(fin_done, fin_scores.min(1, keepdim=True).values, torch.full_like(fin_scores, -1.0e9))
        open_row = open_row & (best_run > worst).any(-1, keepdim=True)
        if not bool(open_row.any()) or bool(hits.all()):
            break
    return fin_seqs[:, 0]


def sequence_bias_(scores, ids, sequence_bias):
    """HF:generation/logits_process.py SequenceBiasLogitsProcessor (FIRST in HF's processor order; the reference passes a table
    through model.generate(sequence_bias=...), evaluation.py:362-364): a length-1 entry adds its bias to its token everywhere;
    a longer entry adds its bias to its LAST token in every row whose history ends with the tokens before it."""
    Bn, cur = ids.shape
    for toks, bias in sequence_bias.items():
        toks = tuple(int(t) for t in toks)
        if len(toks) == 1:
            scores[:, toks[0]] += bias
        elif len(toks) - 1 <= cur:
            pre = torch.tensor(toks[:-1], dtype=ids.dtype)
            hit = (ids[:, cur - len(pre):] == pre).all(1)
            scores[hit, toks[-1]] += bias
    return scores


def sequence_score(sd, x, dims, seq, prompt_len, repetition_penalty=1.0, no_repeat_ngram_size=0, suppress_tokens=(),
                   begin_suppress_tokens=(), length_penalty=1.0, eos_id=None, forced_decoder_ids=None, sequence_bias=None):
    """The score HF's beam search (HF:generation/utils.py:3208-3545; beam_search above) assigns to ONE given hypothesis, in fp32:
    the sum of the PROCESSED log-probabilities of its generated tokens (log_softmax -> sequence bias -> repetition penalty ->
    no-repeat-ngram -> suppress lists -> forced ids, each on the prefix before the token) / generated_length ** length_penalty,
    where a hypothesis ends at its first EOS after the prompt (else at the row's end).  x (1, ch, T), seq (L,) prompt + tokens
    (+ pad).  The GPU tests hold a beam row that left the reference's best hypothesis against this: a valid alternative scores
    within fp16 resolution of the best under the REFERENCE arithmetic, whatever the path under test computed for it."""
    eos_id = dims.eos_id if eos_id is None else eos_id
    seq = torch.as_tensor(seq, dtype=torch.long).view(1, -1)
    P = int(prompt_len)
    gen = seq[0, P:]
    is_eos = (gen == eos_id).nonzero()
    end = P + (int(is_eos[0]) + 1 if len(is_eos) else gen.numel())
    enc = encoder(sd, x, dims)
    hid = decoder(sd, seq[:, :end - 1], enc, dims)
    logits = F.linear(hid[0], sd["model.decoder.embed_tokens.weight"]).float()        # row t predicts token t + 1
    begin = begin_index_for(P, forced_decoder_ids)
    total = 0.0
    for cur in range(P, end):
        lp = F.log_softmax(logits[cur - 1:cur], -1).clone()
        ids = seq[:, :cur]
        if sequence_bias:
            sequence_bias_(lp, ids, sequence_bias)
        if repetition_penalty != 1.0:
            repetition_penalty_(lp, ids, repetition_penalty)
        if no_repeat_ngram_size > 0:
            no_repeat_ngram_(lp, ids, no_repeat_ngram_size)
        suppress_(lp, list(suppress_tokens), list(begin_suppress_tokens), cur, begin)
        force_tokens_(lp, forced_decoder_ids, cur)
        total += float(lp[0, int(seq[0, cur])])
    return total / float(end - P) ** length_penalty


# --------------------------------------------------------------------------- data feed
def reader_pad_sample(sample: np.ndarray, dataset_name: str | None, modal_ch: int, max_len: int) -> np.ndarray:
    """utils/reader.py:272-280 (dataset-specific channel slice, channel zero-pad) + :496-506 (time crop / right zero-pad)."""
    if dataset_name == "schoffelen":
        sample = sample[28:301]
    elif dataset_name == "gwilliams":
        sample = sample[:208]
    else:
        sample = sample[:modal_ch]
    if modal_ch > sample.shape[0]:
        sample = np.pad(sample, ((0, modal_ch - sample.shape[0]), (0, 0)))
    sample = sample[:, :max_len]
    sample = np.pad(sample, ((0, 0), (0, max_len - sample.shape[-1])))
    assert sample.shape == (modal_ch, max_len)
    return sample


def collate(features, pad_id, bos_id):
    """utils/data_utils.py:185-221: stack input_features[0] as float32; right-pad labels, pad -> -100;
    strip a leading BOS only when EVERY row starts with it."""
    x = torch.stack([torch.tensor(f["input_features"][0], dtype=torch.float32) for f in features])
    L = max(len(f["labels"]) for f in features)
    lab = torch.full((len(features), L), -100, dtype=torch.long)
    for i, f in enumerate(features):
        lab[i, :len(f["labels"])] = torch.tensor(f["labels"], dtype=torch.long)
    if bool((lab[:, 0] == bos_id).all()):
        lab = lab[:, 1:]
    return {"input_features": x, "labels": lab}


def match_modules_string(names, start_prefixes, end_suffixes, mid_prefixes=()):
    """utils/load_model.py:48-85: startswith any prefix, (contains any mid-fix), endswith any suffix."""
    out = []
    for n in names:
        if not any(n.startswith(s) for s in start_prefixes):
            continue
        if mid_prefixes and not any(m in n for m in mid_prefixes):
            continue
        if any(n.endswith(s) for s in end_suffixes):
            out.append(n)
    return out
