"""GPU: SURVEY §8 f2 -- a checkpoint directory written by STOCK transformers `save_pretrained` (tests/hf_ckpt.py;
what finetune.py:127-131 / evaluation.py:72-74 load) goes through the build's `from_pretrained`, gets the MEG
front-end installed the way evaluation.py:76-86 does, and must reproduce the logits and the generated ids of the
reference object on the same directory (tests/golden/hf_ckpt_tiny.npz, tools/make_goldens.py hf_ckpt).  `generate`
is called WITHOUT suppress lists or a length: they come from the checkpoint's generation_config.json."""
import os

import numpy as np
import pytest
import torch

from neuspeech1_amd.weights import TINY, synth_batch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def model(dev, tmp_path_factory):
    from tests.hf_ckpt import front_end_state, write_stock_hf_checkpoint
    from utils.load_model import WhisperForConditionalGeneration
    from utils.model_utils import projection_module
    path = str(tmp_path_factory.mktemp("stock_hf"))
    write_stock_hf_checkpoint(TINY, path)
    m = WhisperForConditionalGeneration.from_pretrained(path, device_map="auto", local_files_only=True)
    conv1 = projection_module(config_name="base", meg_ch=TINY.ch, d_model=m.model.encoder.conv2.in_channels)
    conv1.load_state_dict({k: torch.from_numpy(v) for k, v in front_end_state(TINY).items()})
    m.model.encoder.set_input_embeddings(conv1.to(m.device))
    return m


def test_logits_of_a_stock_hf_checkpoint(model):
    g = np.load(os.path.join(G, "hf_ckpt_tiny.npz"))
    x, labels = synth_batch(TINY, int(g["B"]), 1234)
    out = model(input_features=torch.from_numpy(x), labels=torch.from_numpy(labels))
    assert abs(out.loss.item() - float(g["loss"])) < 2e-3 * float(g["loss"])
    lg = out.logits.float().cpu().numpy()
    ref = g["logits"]
    assert np.linalg.norm(lg - ref) / np.linalg.norm(ref) < 1e-2      # fp16 GEMM operands vs the fp32 reference object


@pytest.mark.parametrize("name,nb", [("greedy_rp", 1), ("beam5_rp", 5)])
def test_generate_takes_its_defaults_from_generation_config_json(model, name, nb):
    g = np.load(os.path.join(G, "hf_ckpt_tiny.npz"))
    x, labels = synth_batch(TINY, int(g["B"]), 1234)
    out = model.generate(torch.from_numpy(x), do_sample=False, num_beams=nb, repetition_penalty=5.0, no_repeat_ngram_size=2,
                         decoder_input_ids=torch.from_numpy(labels[:, :4].copy()))
    ref = g[name]
    assert out.shape[1] == ref.shape[1] == model.generation_config.max_length       # max_length came from the file
    assert np.array_equal(out.cpu().numpy(), ref), f"\n{out.cpu().numpy().tolist()}\n{ref.tolist()}"
    sup = set(model.generation_config.suppress_tokens)
    assert not (set(out[:, 4:].cpu().numpy().ravel().tolist()) & sup)


def test_generate_refuses_out_of_range_ids(model):
    x, labels = synth_batch(TINY, 2, 1234)
    for kw in (dict(suppress_tokens=[TINY.vocab]), dict(begin_suppress_tokens=[-1]), dict(forced_decoder_ids=[[1, TINY.vocab + 3]])):
        with pytest.raises(ValueError):
            model.generate(torch.from_numpy(x), num_beams=1, max_new_tokens=4, **kw)
