"""MEG front-end factory + Trainer shims (reference utils/model_utils.py).

`projection_module('base', meg_ch=, d_model=)` returns the same module SHAPE the reference builds
(Sequential(Conv1d(ch->d,k3,p1), GELU, Conv1d(d->d,k3,s2,p1)) with `.stride = (2,)`, state-dict keys
0.weight/0.bias/2.weight/2.bias) so checkpoints interchange; on the HIP path the module is a parameter
container — the arithmetic runs in the fused conv-as-GEMM kernels of libneuspeech_hip."""
import os

import torch
import torch.nn as nn

IGNORE_TOKEN_ID = -100


def projection_module(config_name="", **kwargs):
    if config_name == "base":
        d_model = kwargs["d_model"]
        conv1 = nn.Sequential(
            nn.Conv1d(kwargs["meg_ch"], d_model, kernel_size=3, padding=1),
            nn.GELU(),
            nn.Conv1d(d_model, d_model, kernel_size=3, stride=2, padding=1),
        )
        conv1.stride = (2,)   # lets the stock length check 1500*stride*stride = 6000 hold (reference :17)
        return conv1
    if config_name == "replace":   # reference :18-20: one strided conv straight from the MEG channels
        return nn.Conv1d(kwargs["meg_ch"], kwargs["d_model"], kernel_size=3, stride=2, padding=1)
    raise NotImplementedError(config_name)


def load_from_checkpoint(resume_from_checkpoint, model=None):
    pass


def trainer_save_model(output_dir=None, state_dict=None):
    os.makedirs(output_dir, exist_ok=True)


def compute_accuracy(pred):
    predict_res = torch.as_tensor(pred.predictions[0])
    pred_ids = predict_res.argmax(dim=2)
    labels_actual = torch.as_tensor(pred.label_ids).long()
    acc = torch.sum(torch.all(torch.eq(pred_ids, labels_actual), dim=1)) / labels_actual.shape[0]
    return {"accuracy": acc}
