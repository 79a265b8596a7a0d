"""The reference object, built from third-party parts that ship in the image (never a reference file): stock
`transformers.WhisperForConditionalGeneration` (eager attention, fp32 weights) with the build's own 3-line conv stack
installed through `encoder.set_input_embeddings` -- what /root/reference/evaluation.py:72-86 constructs -- loaded with
the seeded synthetic weights of `neuspeech1_amd.weights`.

Two users, both measurement / checking only (the product path never imports this):
  * bench.py: the CPU leg (`cpu_baseline`, fp32 on the host cores) and the `torch_rocm_reference_object` context leg
    (the same object on the GPU under `torch.autocast('cuda', torch.float16)`, the reference's own numerics:
    /root/reference/evaluation.py:350, finetune.py:242);
  * tests/test_live_fp16_gpu.py: the LIVE fp16-autocast run on the GPU box that SURVEY.md §8c (G2) / Appendix A name the
    authoritative check of the numerics contract.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_reference_object(dims, device="cpu", seed_w: int = 42, train_convs: bool = True):
    import torch
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from neuspeech1_amd.weights import make_state_dict
    from utils.model_utils import projection_module
    cfg = WhisperConfig(vocab_size=dims.vocab, num_mel_bins=80, d_model=dims.d, encoder_layers=dims.enc_layers,
                        decoder_layers=dims.dec_layers, encoder_attention_heads=dims.heads,
                        decoder_attention_heads=dims.heads, encoder_ffn_dim=dims.ffn, decoder_ffn_dim=dims.ffn,
                        max_source_positions=dims.src_pos, max_target_positions=dims.tgt_pos, pad_token_id=dims.pad_id,
                        bos_token_id=dims.bos_id, eos_token_id=dims.eos_id, decoder_start_token_id=dims.start_id,
                        attn_implementation="eager", suppress_tokens=[], begin_suppress_tokens=[])
    model = WhisperForConditionalGeneration(cfg)
    model.model.encoder.set_input_embeddings(projection_module(config_name="base", meg_ch=dims.ch, d_model=dims.d))
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(dims, seed_w).items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("proj_out" in m for m in missing), missing
    for p_ in model.parameters():
        p_.requires_grad_(False)
    if train_convs:     # modules_to_save of the reference (finetune.py:202): the conv stem trains
        for n_, p_ in model.named_parameters():
            if n_.startswith("model.encoder.conv"):
                p_.requires_grad_(True)
    model.to(device)
    model.eval()
    return model


def generate_kwargs(dims, prompt, max_new_tokens, suppress_eos=False):
    """explicit lists / ids (SURVEY.md §8c caveat beta): nothing is taken from a default generation config"""
    return dict(do_sample=False, max_new_tokens=max_new_tokens, decoder_input_ids=prompt,
                suppress_tokens=[dims.eos_id] if suppress_eos else None, begin_suppress_tokens=None,
                pad_token_id=dims.pad_id, eos_token_id=dims.eos_id)


def generate(model, feats, **kw):
    """reference-era call shape (utils/load_model.py:1314-1322 -> super().generate): GenerationMixin.generate, which
    returns prompt + new tokens (SURVEY.md §8c caveat alpha)"""
    import transformers
    return transformers.GenerationMixin.generate(model, feats, **kw)
