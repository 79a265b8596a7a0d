"""`WhisperForConditionalGeneration` with the reference's surface (reference utils/load_model.py:327-1401,
used at finetune.py:18,127-148 and, as the stock HF class, at evaluation.py:72-86) on the MI355X engine.

The nn.Module tree only CARRIES parameters under HuggingFace's dotted names (LoRA target selection is by name,
finetune.py:189-198); forward / backward / generate run in libneuspeech_hip through
neuspeech1_amd.engine.MegWhisperEngine.  There is no CPU fallback: without a GPU + the HIP library, forward raises.
"""
from __future__ import annotations

import json
import os
import re
from types import SimpleNamespace

import torch
from torch import nn

from neuspeech1_amd.weights import TINY, WHISPER_BASE, WHISPER_LARGE_V2, WhisperDims, make_state_dict


# ----------------------------------------------------------------------------- name matchers (reference :48-100)
def match_modules_string(named_modules, start_prefixes, end_suffixes, mid_prefixes=[]):
    """names that start with any prefix, (contain any mid-fix when given) and end with any suffix, in iteration order."""
    out = []
    for name, _ in named_modules:
        if not any(name.startswith(p) for p in start_prefixes):
            continue
        if mid_prefixes and not any(m in name for m in mid_prefixes):
            continue
        if any(name.endswith(s) for s in end_suffixes):
            out.append(name)
    return out


def match_modules(named_modules, prefix_list, suffix_list, mid_fix_list=[""]):
    """regex form ^(prefix).*(mid).*(suffix)$; a name is appended once per matching (prefix, suffix) pair."""
    out = []
    for name, _ in named_modules:
        for prefix in prefix_list:
            for suffix in suffix_list:
                for mid in mid_fix_list:
                    if re.match(re.compile(rf"^({prefix}).*({mid}).*({suffix})$"), name):
                        out.append(name)
                        break
    return out


# ----------------------------------------------------------------------------- parameter-carrying module tree
class WhisperAttention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)


class WhisperEncoderLayer(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.self_attn = WhisperAttention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)
        self.final_layer_norm = nn.LayerNorm(d)


class WhisperDecoderLayer(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.self_attn = WhisperAttention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.encoder_attn = WhisperAttention(d)
        self.encoder_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)
        self.final_layer_norm = nn.LayerNorm(d)


class WhisperEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.d_model
        self.conv1 = nn.Conv1d(cfg.num_mel_bins, d, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(d, d, kernel_size=3, stride=2, padding=1)
        self.embed_positions = nn.Embedding(cfg.max_source_positions, d)
        self.layers = nn.ModuleList([WhisperEncoderLayer(d, cfg.encoder_ffn_dim) for _ in range(cfg.encoder_layers)])
        self.layer_norm = nn.LayerNorm(d)

    def get_input_embeddings(self):
        return self.conv1

    def set_input_embeddings(self, value: nn.Module):
        self.conv1 = value


class WhisperDecoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self._init_std = getattr(cfg, "init_std", 0.02)
        d = cfg.d_model
        self.embed_tokens = nn.Embedding(cfg.vocab_size, d)
        self.embed_positions = nn.Embedding(cfg.max_target_positions, d)
        self.layers = nn.ModuleList([WhisperDecoderLayer(d, cfg.decoder_ffn_dim) for _ in range(cfg.decoder_layers)])
        self.layer_norm = nn.LayerNorm(d)

    def post_init(self):
        _hf_init_weights(self, self._init_std)


def _sinusoids(length, channels, max_timescale=10000.0):
    """HF:modeling_whisper.py sinusoids(): the encoder's fixed position table"""
    import math
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
    return torch.cat([t.sin(), t.cos()], dim=1)


@torch.no_grad()
def _hf_init_weights(root: nn.Module, std: float):
    """transformers PreTrainedModel.post_init() -> init_weights(): what --random_initialize_whisper triggers in the
    reference (finetune.py:166-168 on the whole model, evaluation.py:90-91 on the decoder).  Linear / Conv1d / Embedding
    weights ~ N(0, init_std), biases 0, LayerNorm (1, 0), the encoder's position table back to its sinusoids."""
    for m in root.modules():
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            m.weight.normal_(0.0, std)
            if m.bias is not None:
                m.bias.zero_()
        elif isinstance(m, nn.Embedding):
            m.weight.normal_(0.0, std)
        elif isinstance(m, nn.LayerNorm):
            m.weight.fill_(1.0)
            m.bias.zero_()
    for m in root.modules():
        if isinstance(m, WhisperEncoder):
            w = m.embed_positions.weight
            w.copy_(_sinusoids(*w.shape).to(w))


class WhisperModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.encoder = WhisperEncoder(cfg)
        self.decoder = WhisperDecoder(cfg)


_SYNTH = {"tiny": TINY, "base": WHISPER_BASE, "large-v2": WHISPER_LARGE_V2}


def _cfg_from_dims(dm: WhisperDims) -> SimpleNamespace:
    return SimpleNamespace(
        vocab_size=dm.vocab, num_mel_bins=80, d_model=dm.d, encoder_layers=dm.enc_layers, decoder_layers=dm.dec_layers,
        encoder_attention_heads=dm.heads, decoder_attention_heads=dm.heads, encoder_ffn_dim=dm.ffn,
        decoder_ffn_dim=dm.ffn, max_source_positions=dm.src_pos, max_target_positions=dm.tgt_pos,
        pad_token_id=dm.pad_id, bos_token_id=dm.bos_id, eos_token_id=dm.eos_id, decoder_start_token_id=dm.start_id,
        forced_decoder_ids=None, suppress_tokens=[], begin_suppress_tokens=[], use_cache=True, max_length=dm.tgt_pos)


class Seq2SeqLMOutput(dict):
    __getattr__ = dict.get


class WhisperForConditionalGeneration(nn.Module):
    base_model_prefix = "model"

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = WhisperModel(config)
        self.proj_out = nn.Linear(config.d_model, config.vocab_size, bias=False)
        self.proj_out.weight = self.model.decoder.embed_tokens.weight    # tied (reference :947)
        self._engine = None
        self._peft = None          # set by neuspeech1_amd.peft_compat.get_peft_model
        self.train_cfg = None
        self.generation_config = None   # SimpleNamespace of generation_config.json when the checkpoint carries one
        self.train(False)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, path, load_in_8bit=False, device_map=None, local_files_only=True, **_):
        """`path`: a directory with config.json + model.safetensors (HF names), or `synthetic:{tiny|base|large-v2}[:seed]`
        for the documented seeded random init (no pretrained weights exist offline)."""
        if load_in_8bit:
            raise NotImplementedError("8-bit loading is outside the MI355X hot path")
        if isinstance(path, str) and path.startswith("synthetic:"):
            parts = path.split(":")
            dm = _SYNTH[parts[1]]
            seed = int(parts[2]) if len(parts) > 2 else 42
            cfg0 = _cfg_from_dims(dm)
            cfg0.synthetic_name = parts[1]       # travels in config.json: lets tools rebuild the synthetic processor
            model = cls(cfg0)
            sd = {k: torch.from_numpy(v) for k, v in make_state_dict(dm, seed).items() if "conv1." not in k}
            model.load_state_dict(sd, strict=False)
        else:
            with open(os.path.join(path, "config.json")) as f:
                raw = json.load(f)
            cfg = _cfg_from_dims(WHISPER_BASE)
            for k, v in raw.items():
                setattr(cfg, k, v)
            model = cls(cfg)
            from safetensors.torch import load_file
            sd = load_file(os.path.join(path, "model.safetensors"))
            # a merged export (merge_lora.py) carries its MEG front-end: restore the module so its weights load
            e = "model.encoder."
            if e + "conv1.0.weight" in sd or (e + "conv1.weight" in sd and sd[e + "conv1.weight"].shape[1] != cfg.num_mel_bins):
                from utils.model_utils import projection_module
                w = sd.get(e + "conv1.0.weight", sd.get(e + "conv1.weight"))
                model.model.encoder.set_input_embeddings(projection_module(
                    config_name="base" if e + "conv1.0.weight" in sd else "replace", meg_ch=w.shape[1], d_model=cfg.d_model))
            own = model.state_dict()
            sd = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
            model.load_state_dict(sd, strict=False)
            # generation defaults travel beside the weights (hub whisper: max_length 448, the suppress lists,
            # forced_decoder_ids [[1, null], [2, 50359]]).  evaluation.py:72-74 loads a fresh model, so
            # these apply at decode.
            gpath = os.path.join(path, "generation_config.json")
            if os.path.exists(gpath):
                with open(gpath) as f:
                    model.generation_config = SimpleNamespace(**json.load(f))
        dev = _resolve_device(device_map)
        return model.to(dev)

    def save_pretrained(self, save_directory, **_):
        os.makedirs(save_directory, exist_ok=True)
        cfg = {k: v for k, v in vars(self.config).items() if isinstance(v, (int, float, str, list, type(None), bool))}
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(cfg, f, indent=1)
        from safetensors.torch import save_file
        sd = {k: v.detach().contiguous().cpu() for k, v in self.state_dict().items() if k != "proj_out.weight"}
        save_file(sd, os.path.join(save_directory, "model.safetensors"))
        if self.generation_config is not None:      # a merged export keeps the base checkpoint's generation defaults
            with open(os.path.join(save_directory, "generation_config.json"), "w") as f:
                json.dump(vars(self.generation_config), f, indent=1)

    # ------------------------------------------------------------------ HF-style accessors
    @property
    def device(self):
        return self.model.decoder.embed_tokens.weight.device

    def get_encoder(self):
        return self.model.encoder

    def get_decoder(self):
        return self.model.decoder

    def get_input_embeddings(self):
        return self.model.decoder.embed_tokens

    def get_output_embeddings(self):
        return self.proj_out

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def invalidate_engine(self):
        """call after editing frozen weights in place"""
        self._engine = None
        self.release_decode_sessions()

    def release_decode_sessions(self):
        """Free what generate() keeps between calls: the Generator of this model holds up to two decode SESSIONS (KV caches, cross K|V and
        transposed-V images, recorded launch lists, hipGraph pools: up to a quarter of the device memory; ~62 GB each at large-v2, B = 128,
        beam 5).  evaluation.py calls this when its loop ends; train() calls it when the model goes back to training; a caller that
        evaluates and then trains, or loads a second model, in the same process gets the memory back."""
        gen = getattr(self, "_generator", None)
        if gen is not None:
            gen.clear_sessions()
        self._generator = None

    def train(self, mode: bool = True):
        if mode:
            self.release_decode_sessions()
        return super().train(mode)

    def post_init(self):
        """random re-initialisation of every weight (what --random_initialize_whisper does through HF's post_init)"""
        _hf_init_weights(self, getattr(self.config, "init_std", 0.02))
        self._engine = None

    # ------------------------------------------------------------------ engine
    def dims(self) -> WhisperDims:
        c, conv = self.config, self.model.encoder.conv1
        first = conv[0] if isinstance(conv, nn.Sequential) else conv
        return WhisperDims(d=c.d_model, heads=c.encoder_attention_heads, ffn=c.encoder_ffn_dim,
                           enc_layers=c.encoder_layers, dec_layers=c.decoder_layers, vocab=c.vocab_size,
                           src_pos=c.max_source_positions, tgt_pos=c.max_target_positions, ch=first.in_channels,
                           pad_id=c.pad_token_id, bos_id=c.bos_token_id, eos_id=c.eos_token_id,
                           start_id=c.decoder_start_token_id)

    def engine(self):
        """Build (once) the HIP engine from the current parameters; trainable parameters are re-pointed at the
        engine's flat fp32 buffer so optimizer updates and state_dict() see the same storage."""
        if self._engine is not None:
            return self._engine
        from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
        if self.device.type != "cuda":
            raise RuntimeError("the MI355X engine needs the model on a GPU (no CPU fallback for the product path)")
        conv = self.model.encoder.conv1
        if not isinstance(conv, nn.Sequential) and conv.stride != (2,):
            raise RuntimeError("install the MEG front-end first: model.model.encoder.set_input_embeddings("
                               "projection_module('base' | 'replace', meg_ch=..., d_model=...))")
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        lora, lora_sd = None, None
        if self._peft is not None:
            pc = self._peft
            ada = getattr(pc, "peft_type", "LORA") == "ADALORA"
            from neuspeech1_amd.peft_compat import targets_cover_decoder
            full = targets_cover_decoder(self, pc.target_modules)      # --ft_full: decoder projections adapted too
            lora = LoraSpec(r=pc.init_r if ada else pc.r, alpha=float(pc.lora_alpha), dropout=float(pc.lora_dropout),
                            adalora=ada, orth_reg_weight=float(getattr(pc, "orth_reg_weight", 0.0)) if ada else 0.0,
                            layers=None if full else len(pc.target_modules) // 6, decoder=full)
            lora_sd = {}
            for k, v in sd.items():
                if ".lora_" in k:
                    k = k.replace(".default", "")
                    lora_sd[k if k.endswith(".weight") else k + ".weight"] = v
        eng = MegWhisperEngine(self.dims(), sd, lora=lora, lora_sd=lora_sd, train_cfg=self.train_cfg or TrainCfg(),
                               device=self.device)
        self._engine = eng
        self._tie_trainables(eng)
        return eng

    def _tie_trainables(self, eng):
        enc = self.model.encoder
        for name, mod in self._conv_modules():
            mod.weight.data = eng.conv_weight(name)
            mod.bias.data = eng.pview(f"model.encoder.{name}.bias")
        if self._peft is None:
            return
        for par, view in self._adapter_views(eng, eng.pview):
            par.data = view

    def _conv_modules(self):
        """(engine name, module) of the trainable convs: 'base' front-end = Sequential(conv, GELU, conv), 'replace' = one conv"""
        enc = self.model.encoder
        if isinstance(enc.conv1, nn.Sequential):
            return (("conv1.0", enc.conv1[0]), ("conv1.2", enc.conv1[2]), ("conv2", enc.conv2))
        return (("conv1", enc.conv1), ("conv2", enc.conv2))

    def _adapter_views(self, eng, getv):
        """(parameter, view of the engine's flat buffer) for every adapter tensor.  The engine pads the rank to a
        multiple of 16 (eng.r) and keeps q|k|v down-projections stacked; the parameters see the live r rows/columns."""
        d, f, rp, r = eng.dims.d, eng.dims.ffn, eng.r, eng.r_real
        ada = eng.adalora
        W = (lambda m, t: getattr(m, t)["default"]) if ada else (lambda m, t: getattr(m, t)["default"].weight)
        for i, layer in enumerate(self.model.encoder.layers[:eng.n_lora]):
            p = f"model.encoder.layers.{i}."
            A = getv(p + "self_attn.qkv.lora_A").view(3, rp, d)
            E = getv(p + "self_attn.qkv.lora_E").view(3, rp) if ada else None
            for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                m = getattr(layer.self_attn, nm)
                yield W(m, "lora_A"), A[j, :r]
                yield W(m, "lora_B"), getv(p + f"self_attn.{nm}.lora_B").view(d, rp)[:, :r]
                if ada:
                    yield W(m, "lora_E"), E[j, :r].unsqueeze(1)
            for nm, m, no, ki in (("self_attn.out_proj", layer.self_attn.out_proj, d, d), ("fc1", layer.fc1, f, d),
                                  ("fc2", layer.fc2, d, f)):
                yield W(m, "lora_A"), getv(p + nm + ".lora_A").view(rp, ki)[:r]
                yield W(m, "lora_B"), getv(p + nm + ".lora_B").view(no, rp)[:, :r]
                if ada:
                    yield W(m, "lora_E"), getv(p + nm + ".lora_E")[:r].unsqueeze(1)

        if not eng.dec_lora:
            return
        from neuspeech1_amd.engine import _dec_sites
        for i, layer in enumerate(self.model.decoder.layers):
            p = f"model.decoder.layers.{i}."
            for site, _, projs, kin, nout, _ in _dec_sites(d, f):
                G = len(projs)
                A = getv(p + site + ".lora_A").view(G, rp, kin)
                E = getv(p + site + ".lora_E").view(G, rp) if ada else None
                for j, pj in enumerate(projs):
                    m = layer.get_submodule(pj)
                    yield W(m, "lora_A"), A[j, :r]
                    yield W(m, "lora_B"), getv(p + pj + ".lora_B").view(nout, rp)[:, :r]
                    if ada:
                        yield W(m, "lora_E"), E[j, :r].unsqueeze(1)

    # ------------------------------------------------------------------ forward / generate
    def forward(self, input_features=None, attention_mask=None, decoder_input_ids=None, labels=None, **_):
        """reference utils/load_model.py:976-1070: returns an object with .loss (0-d) and .logits (B, L, V)."""
        eng = self.engine()
        eng.refresh_operands()   # trainables may have been stepped by an external torch optimizer
        x = input_features.to(self.device, torch.float32).contiguous()
        if labels is not None:
            labels = labels.to(self.device)
        if decoder_input_ids is not None:
            decoder_input_ids = decoder_input_ids.to(self.device)
        want_grad = self.training and torch.is_grad_enabled() and labels is not None
        if want_grad:
            loss = _TrainStepFn.apply(self, x, labels, *self._trainable_tensors())
            logits = None
        else:
            loss, logits = eng.forward(x, labels, decoder_input_ids=decoder_input_ids, train=False)
            loss = loss.clone().reshape(()) if loss is not None else None
            logits = logits.clone()
        return Seq2SeqLMOutput(loss=loss, logits=logits)

    def _trainable_tensors(self):
        return [p for p in self.parameters() if p.requires_grad]

    @torch.no_grad()
    def generate(self, input_features=None, inputs=None, do_sample=False, num_beams=1, repetition_penalty=1.0,
                 no_repeat_ngram_size=0, decoder_input_ids=None, max_new_tokens=None, max_length=None,
                 length_penalty=1.0, suppress_tokens=None, begin_suppress_tokens=None, eos_token_id=None,
                 pad_token_id=None, sequence_bias=None, forced_decoder_ids=None, **_):
        """evaluation.py:370-386 call shape; returns prompt + generated ids (B, <= max_length) int64."""
        if do_sample:
            raise NotImplementedError("sampling is outside the hot path (the reference decodes with do_sample=False)")
        from neuspeech1_amd.generate import Generator
        eng = self.engine()
        eng.refresh_operands()
        x = (input_features if input_features is not None else inputs).to(self.device, torch.float32).contiguous()
        c, gc = self.config, self.generation_config

        def default(name, explicit=None, fallback=None):
            """explicit argument > generation_config.json > model config (HF: the generation config rules at generate;
            one derived from the model config stands in when the checkpoint has no generation_config.json)"""
            if explicit is not None:
                return explicit
            for src in (gc, c):
                v = getattr(src, name, None) if src is not None else None
                if v is not None:
                    return v
            return fallback

        def one(v):
            # HF treats EVERY id of a list-valued eos_token_id as end-of-sequence; the decode kernels take one id.  The
            # openai/whisper-* generation configs carry a single id (50257); a checkpoint listing several would silently
            # decode past the others, so refuse it rather than collapse it to its first element
            if isinstance(v, (list, tuple)):
                if len(set(v)) > 1:
                    raise NotImplementedError(f"generate: several end-of-sequence / pad ids {list(v)} (the HIP decode loop stops on one id)")
                return v[0]
            return v

        if decoder_input_ids is None:
            decoder_input_ids = torch.full((x.shape[0], 1), default("decoder_start_token_id"), dtype=torch.int64)
        prompt = decoder_input_ids.to(self.device, torch.int64).contiguous()
        P = prompt.shape[1]
        limit = default("max_length", max_length, c.max_target_positions)
        new = max_new_tokens if max_new_tokens is not None else limit - P
        sup = default("suppress_tokens", suppress_tokens, [])
        bsup = default("begin_suppress_tokens", begin_suppress_tokens, [])
        # forced_decoder_ids, resolved like the reference's wrapper (utils/load_model.py:1210-1222: model config first,
        # then the generation config, then the keyword) and applied like HF's ForceTokensLogitsProcessor; the
        # begin-suppress list then applies after the last forced position (HF generation/utils.py of the reference's era:
        # begin_index = prompt length + forced_decoder_ids[-1][0])
        forced = getattr(c, "forced_decoder_ids", None)
        if forced is None and gc is not None:
            forced = getattr(gc, "forced_decoder_ids", None)
        if forced is None:
            forced = forced_decoder_ids
        begin_index = P + int(forced[-1][0]) if forced else P
        # one Generator per engine: it keeps the decode sessions (device state + recorded launch lists / captured hipGraphs per call
        # signature), so that the evaluation loop's second batch captures and every later one only replays (neuspeech1_amd/generate.py)
        gen = getattr(self, "_generator", None)
        if gen is None or gen.eng is not eng:
            gen = self._generator = Generator(eng)
        return gen.generate(x, prompt, num_beams=num_beams, max_new_tokens=new,
                            repetition_penalty=repetition_penalty, no_repeat_ngram_size=no_repeat_ngram_size,
                            suppress_tokens=list(sup), begin_suppress_tokens=list(bsup),
                            length_penalty=length_penalty, eos_id=one(default("eos_token_id", eos_token_id)),
                            pad_id=one(default("pad_token_id", pad_token_id)), sequence_bias=sequence_bias,
                            forced_decoder_ids=forced, begin_index=begin_index)


def _resolve_device(device_map):
    if device_map in (None, "auto", "cuda"):
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if isinstance(device_map, dict):
        return torch.device("cuda", int(device_map[""]))
    return torch.device(device_map)


class _TrainStepFn(torch.autograd.Function):
    """`.loss.backward()` compatibility: one autograd node around the engine's explicit forward / backward.
    Gradients of the trainable parameters are read out of the engine's flat gradient buffer (unscaled)."""

    @staticmethod
    def forward(ctx, model, x, labels, *params):
        eng = model.engine()
        eng.zero_grad()
        loss, _ = eng.forward(x, labels, train=True, compute_grad=True)
        ctx.model, ctx.params = model, params
        return loss.clone().reshape(())

    @staticmethod
    def backward(ctx, g):
        model, eng = ctx.model, ctx.model.engine()
        eng.backward()
        scale = g / eng.loss_scale_dev
        grads = []
        gmap = model._grad_views(eng)
        for p in ctx.params:
            v = gmap.get(p.data_ptr())
            grads.append(None if v is None else (v * scale))
        return (None, None, None, *grads)


def _grad_views(self, eng):
    """parameter storage pointer -> view of the engine gradient buffer with the parameter's shape"""
    out = {}
    enc = self.model.encoder
    for name, mod in self._conv_modules():
        out[mod.weight.data_ptr()] = eng.conv_weight_grad(name)
        out[mod.bias.data_ptr()] = eng.gview(f"model.encoder.{name}.bias")
    if self._peft is not None:
        for par, view in self._adapter_views(eng, eng.gview):
            out[par.data_ptr()] = view
    return out


WhisperForConditionalGeneration._grad_views = _grad_views
