"""The slice of `peft` the reference uses (finetune.py:13,153-154,171,185,205-212; evaluation.py:88-89;
merge_lora.py:43-44), re-stated for the HIP engine: LoraConfig, get_peft_model, PeftModel.from_pretrained /
save_pretrained / merge_and_unload, prepare_model_for_kbit_training.

`peft` itself is not importable offline, so these semantics are the build's DEFINITION:
  y = W x + b + (alpha/r) * B(A(dropout_p(x))),  A ~ kaiming_uniform(a=sqrt(5)), B = 0,
  modules_to_save = trainable copies of encoder.conv1 / encoder.conv2,
  merge: W <- W + (alpha/r) B A.
Adapter files use PEFT's layout: adapter_config.json + adapter_model.safetensors with keys
`base_model.model.<module>.lora_A.weight` / `.lora_B.weight` and `base_model.model.model.encoder.conv1.0.weight` ...
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import asdict, dataclass, field

import torch
from torch import nn

from .weights import LORA_SUFFIXES


@dataclass
class LoraConfig:
    r: int = 8
    lora_alpha: float = 8
    target_modules: list = field(default_factory=list)
    lora_dropout: float = 0.0
    bias: str = "none"
    modules_to_save: list = field(default_factory=list)
    peft_type: str = "LORA"


@dataclass
class AdaLoraConfig(LoraConfig):
    """finetune.py:206-208.  The reference trains with the stock Seq2SeqTrainer and never calls
    `update_and_allocate`, so the rank budget (target_r, tinit, tfinal, deltaT, beta1/2) is never applied: every
    module stays at init_r and AdaLoRA reduces to the SVD-form adapter  y += B((A x) * E) * alpha / (init_r + 1e-5)
    plus the orthogonality penalty  orth_reg_weight * mean(||A A^T - I||_F, ||B^T B - I||_F)  added to the loss
    (peft AdaLoraModel.forward).  The allocator fields are carried for config round-trips only."""
    init_r: int = 12
    target_r: int = 8
    beta1: float = 0.85
    beta2: float = 0.85
    tinit: int = 0
    tfinal: int = 0
    deltaT: int = 1
    orth_reg_weight: float = 0.5
    total_step: int | None = None
    peft_type: str = "ADALORA"


def prepare_model_for_kbit_training(model):
    """peft freezes every parameter and upcasts half params; the engine keeps fp32 masters already."""
    for p in model.parameters():
        p.requires_grad = False
    return model


def _expected_targets(model):
    names = []
    for i in range(len(model.model.encoder.layers)):
        for suf in LORA_SUFFIXES:
            names.append(f"model.encoder.layers.{i}." + (suf if suf.startswith("fc") else f"self_attn.{suf}"))
    return names


def _decoder_targets(model):
    """the ten adapted Linear modules of every decoder layer with --ft_full (finetune.py:191-192, prefixes = ['model'])"""
    names = []
    for i in range(len(model.model.decoder.layers)):
        p = f"model.decoder.layers.{i}."
        for suf in LORA_SUFFIXES:
            names += [p + suf] if suf.startswith("fc") else [p + f"self_attn.{suf}", p + f"encoder_attn.{suf}"]
    return names


def targets_cover_decoder(model, target_modules) -> bool:
    return set(target_modules) == set(_expected_targets(model)) | set(_decoder_targets(model))


class _Base(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model


class PeftModel(nn.Module):
    def __init__(self, model, config: LoraConfig, _inject=True):
        super().__init__()
        self.base_model = _Base(model)
        self.peft_config = {"default": config}
        if _inject:
            self._inject(model, config)

    # -- construction
    @staticmethod
    def _inject(model, config: LoraConfig):
        full = _expected_targets(model)
        n = len(config.target_modules)
        # all six Linear modules of the first N encoder layers (N = all: finetune.py:194, N < all: --fine_tune_layers :189-190)
        # or every encoder AND decoder projection (--ft_full, :191-192)
        if not targets_cover_decoder(model, config.target_modules) and \
                (n == 0 or n % len(LORA_SUFFIXES) or set(config.target_modules) != set(full[:n])):
            raise NotImplementedError("the HIP engine carries adapters on q/k/v/out/fc1/fc2 of the first N encoder layers "
                                      f"(N = 1..{len(full) // len(LORA_SUFFIXES)}) or of the whole model (--ft_full); got {n} "
                                      "target modules that are neither")
        ada = isinstance(config, AdaLoraConfig)
        if not ada and config.r % 16:
            raise NotImplementedError("LoRA rank must be a multiple of 16 (MFMA K granularity)")
        dev = model.device
        for name in config.target_modules:
            lin = model.get_submodule(name)
            if ada:   # peft SVDLinear: parameters (not Linears); A, B ~ N(0, 0.02), E = 0
                r = config.init_r
                lin.lora_A = nn.ParameterDict({"default": nn.Parameter(torch.randn(r, lin.in_features, device=dev) * 0.02)})
                lin.lora_E = nn.ParameterDict({"default": nn.Parameter(torch.zeros(r, 1, device=dev))})
                lin.lora_B = nn.ParameterDict({"default": nn.Parameter(torch.randn(lin.out_features, r, device=dev) * 0.02)})
            else:
                a = nn.Linear(lin.in_features, config.r, bias=False, device=dev)
                b = nn.Linear(config.r, lin.out_features, bias=False, device=dev)
                nn.init.kaiming_uniform_(a.weight, a=math.sqrt(5))
                nn.init.zeros_(b.weight)
                lin.lora_A = nn.ModuleDict({"default": a})
                lin.lora_B = nn.ModuleDict({"default": b})
            lin.weight.requires_grad = False
            if lin.bias is not None:
                lin.bias.requires_grad = False
        for name in config.modules_to_save or []:
            for p in model.get_submodule(name).parameters():
                p.requires_grad = True
        model._peft = config
        model._engine = None

    @classmethod
    def from_pretrained(cls, model, path, is_trainable=False, local_files_only=True, **_):
        with open(os.path.join(path, "adapter_config.json")) as f:
            raw = json.load(f)
        common = dict(lora_alpha=raw["lora_alpha"], target_modules=raw["target_modules"],
                      lora_dropout=raw.get("lora_dropout", 0.0), bias=raw.get("bias", "none"),
                      modules_to_save=raw.get("modules_to_save") or [])
        if raw.get("peft_type") == "ADALORA":
            cfg = AdaLoraConfig(r=raw.get("r", 8), **common,
                                **{k: raw[k] for k in ("init_r", "target_r", "beta1", "beta2", "tinit", "tfinal", "deltaT",
                                                       "orth_reg_weight", "total_step") if k in raw})
        else:
            cfg = LoraConfig(r=raw["r"], **common)
        pm = cls(model, cfg)
        from safetensors.torch import load_file
        sd = load_file(os.path.join(path, "adapter_model.safetensors"))
        own = dict(pm.named_parameters())
        for k, v in sd.items():
            if k.endswith(".ranknum"):
                continue
            kk = k
            for t in ("lora_A", "lora_B", "lora_E"):
                kk = kk.replace(f".{t}.weight", f".{t}.default.weight")
                if kk.endswith("." + t):
                    kk += ".default"
            own[kk].data.copy_(v.to(own[kk].device))
        if not is_trainable:
            for p in pm.parameters():
                p.requires_grad = False
        model._engine = None
        return pm

    def save_pretrained(self, path, **_):
        os.makedirs(path, exist_ok=True)
        cfg = self.peft_config["default"]
        with open(os.path.join(path, "adapter_config.json"), "w") as f:
            json.dump(asdict(cfg), f, indent=1)
        sd = {}
        for k, v in self.named_parameters():
            if ".lora_" in k:
                sd[k.replace(".default", "")] = v.detach().contiguous().cpu()
            elif any(k.startswith("base_model.model." + m + ".") for m in (cfg.modules_to_save or [])):
                sd[k] = v.detach().contiguous().cpu()
        from safetensors.torch import save_file
        save_file(sd, os.path.join(path, "adapter_model.safetensors"))

    # -- use
    @property
    def model(self):
        return self.base_model.model

    @property
    def config(self):
        return self.model.config

    @property
    def device(self):
        return self.model.device

    def forward(self, *a, **k):
        return self.model(*a, **k)

    def generate(self, *a, **k):
        return self.model.generate(*a, **k)

    def engine(self):
        return self.model.engine()

    def print_trainable_parameters(self):
        tr = sum(p.numel() for p in self.parameters() if p.requires_grad)
        al = sum(p.numel() for p in self.parameters())
        print(f"trainable params: {tr:,d} || all params: {al:,d} || trainable%: {100 * tr / al:.4f}")

    def merge_and_unload(self):
        """W <- W + (alpha/r) B A  (AdaLoRA: alpha/(r+1e-5) B (A*E)) on every adapted Linear, adapters removed."""
        model = self.model
        cfg = self.peft_config["default"]
        ada = isinstance(cfg, AdaLoraConfig)
        scale = cfg.lora_alpha / (cfg.init_r + 1e-5) if ada else cfg.lora_alpha / cfg.r
        with torch.no_grad():
            for name in cfg.target_modules:
                lin = model.get_submodule(name)
                if ada:
                    lin.weight.add_(scale * lin.lora_B["default"].float() @ (lin.lora_A["default"].float() * lin.lora_E["default"].float()))
                    del lin.lora_A, lin.lora_B, lin.lora_E
                else:
                    lin.weight.add_(scale * lin.lora_B["default"].weight.float() @ lin.lora_A["default"].weight.float())
                    del lin.lora_A, lin.lora_B
        model._peft = None
        model._engine = None
        return model


def get_peft_model(model, config: LoraConfig) -> PeftModel:
    return PeftModel(model, config)
