"""Test helper: write a checkpoint directory with STOCK transformers `save_pretrained` (config.json +
generation_config.json + model.safetensors) from the documented seeded weights -- the kind of directory
/root/reference/finetune.py:127-131 and evaluation.py:72-74 load (`openai/whisper-*`).  transformers is part of the
image (it is not a reference file); nothing here touches /root/reference.  Used by tools/make_goldens.py (hf_ckpt
golden) and by the from_pretrained tests."""
import numpy as np
import torch

from neuspeech1_amd.weights import WhisperDims, make_state_dict


def GEN_CFG(dims: WhisperDims) -> dict:
    """what the hub's generation_config.json carries for whisper, scaled to the test vocabulary"""
    V = dims.vocab
    return {"suppress_tokens": [1, 2, 7, 34, 630, V - 6, V - 4], "begin_suppress_tokens": [220, dims.eos_id],
            "max_length": 28}


def front_end_state(dims: WhisperDims, seed: int = 42) -> dict:
    """state dict of projection_module('base', ...) (keys 0.* / 2.*) from the seeded generator"""
    sd = make_state_dict(dims, seed)
    e = "model.encoder.conv1."
    return {k[len(e):]: v for k, v in sd.items() if k.startswith(e)}


def write_stock_hf_checkpoint(dims: WhisperDims, path: str, seed: int = 42) -> dict:
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    gc = GEN_CFG(dims)
    cfg = WhisperConfig(vocab_size=dims.vocab, num_mel_bins=80, d_model=dims.d, encoder_layers=dims.enc_layers,
                        decoder_layers=dims.dec_layers, encoder_attention_heads=dims.heads,
                        decoder_attention_heads=dims.heads, encoder_ffn_dim=dims.ffn, decoder_ffn_dim=dims.ffn,
                        max_source_positions=dims.src_pos, max_target_positions=dims.tgt_pos,
                        pad_token_id=dims.pad_id, bos_token_id=dims.bos_id, eos_token_id=dims.eos_id,
                        decoder_start_token_id=dims.start_id, suppress_tokens=gc["suppress_tokens"],
                        begin_suppress_tokens=gc["begin_suppress_tokens"], max_length=gc["max_length"])
    torch.manual_seed(0)
    model = WhisperForConditionalGeneration(cfg)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in make_state_dict(dims, seed).items() if "conv1." not in k}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("conv1" in m for m in missing), missing        # only the stock 80-mel conv1 keeps its random init
    for k, v in gc.items():
        setattr(model.generation_config, k, v)
    model.save_pretrained(path)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}
