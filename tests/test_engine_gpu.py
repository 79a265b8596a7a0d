"""Whole-path parity on the GPU: the HIP engine (through the C ABI) against the CPU oracle on the same
seeded inputs, and against the golden vectors of the reference object (tests/golden, tools/make_goldens.py).

Tolerances (fp16 GEMM operands / fp16-rounded Linear outputs vs an fp32 oracle):
  loss          |d| <= 2e-3 * max(1, loss)
  logits / enc  relative Frobenius error <= 1e-2, and element-wise atol 3e-2 + rtol 3e-2
  gradients     relative Frobenius error <= 3e-2 per tensor (fp16 activations-gradients, loss-scaled)
"""
import os

import numpy as np
import pytest
import torch

from neuspeech1_amd.weights import TINY, WHISPER_BASE, make_lora_state, make_state_dict, synth_batch  # noqa: F401

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def make_engine(dims, dev, lora_r=0, **kw):
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    sd = make_state_dict(dims, 42)
    lora_sd = make_lora_state(dims, lora_r) if lora_r else None
    spec = LoraSpec(r=lora_r, alpha=2.0 * lora_r, dropout=0.0) if lora_r else None
    eng = MegWhisperEngine(dims, sd, lora=spec, lora_sd=lora_sd, train_cfg=TrainCfg(**kw), device=dev)
    return eng, sd, lora_sd


def engine_grads(eng, dims, lora_r):
    """Engine gradient buffer -> oracle-named fp32 tensors (unscaled)."""
    s = eng.loss_scale_dev.item()
    out = {}
    for nm in ("conv1.0", "conv1.2", "conv2"):
        out[f"model.encoder.{nm}.weight"] = eng.conv_weight_grad(nm).float().cpu() / s
        out[f"model.encoder.{nm}.bias"] = eng.gview(f"model.encoder.{nm}.bias").float().cpu() / s
    if lora_r:
        d, f, r = dims.d, dims.ffn, lora_r
        for i in range(dims.enc_layers):
            p = f"model.encoder.layers.{i}."
            A = eng.gview(p + "self_attn.qkv.lora_A").view(3, r, d).cpu() / s
            for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                out[p + f"self_attn.{nm}.lora_A.weight"] = A[j]
                out[p + f"self_attn.{nm}.lora_B.weight"] = eng.gview(p + f"self_attn.{nm}.lora_B").view(d, r).cpu() / s
            for nm, no, ki in (("self_attn.out_proj", d, d), ("fc1", f, d), ("fc2", d, f)):
                out[p + nm + ".lora_A.weight"] = eng.gview(p + nm + ".lora_A").view(r, ki).cpu() / s
                out[p + nm + ".lora_B.weight"] = eng.gview(p + nm + ".lora_B").view(no, r).cpu() / s
    return out


@pytest.mark.parametrize("lora_r", [0, 32])
def test_tiny_forward_backward_vs_oracle(dev, lora_r):
    from oracle import whisper_meg_oracle as O
    dims = TINY
    eng, sd, lora_sd = make_engine(dims, dev, lora_r)
    x, labels = synth_batch(dims, 3, 77)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, logits = eng.forward(xd, ld, train=True, compute_grad=False)
    loss_v = loss.item()
    logits_v = logits.float().cpu()
    enc_v = eng._b["enc16"].float().cpu().view(3, dims.src_pos, dims.d)
    o_loss, o_logits, o_enc, o_grads = O.loss_and_grads(sd, lora_sd, x, labels, dims, 2.0 if lora_r else 0.0)
    assert abs(loss_v - o_loss.item()) <= 2e-3 * max(1.0, o_loss.item()), (loss_v, o_loss.item())
    assert rel(enc_v, o_enc) < 1e-2, rel(enc_v, o_enc)
    assert rel(logits_v, o_logits) < 1e-2, rel(logits_v, o_logits)
    torch.testing.assert_close(logits_v, o_logits, atol=3e-2, rtol=3e-2)
    # backward
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    got = engine_grads(eng, dims, lora_r)
    worst = {}
    for k, ref in o_grads.items():
        worst[k] = rel(got[k], ref)
    bad = {k: v for k, v in worst.items() if not v < 3e-2}
    assert not bad, f"gradient mismatch: {bad}"


def test_tiny_matches_reference_golden(dev):
    g = np.load(os.path.join(G, "train_tiny.npz"))
    dims = TINY
    eng, sd, _ = make_engine(dims, dev, 0)
    x, labels = synth_batch(dims, int(g["B"]), int(g["seed_d"]))
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, logits = eng.forward(xd, ld, train=True, compute_grad=True)
    lg = logits.float().cpu().numpy()
    eng.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-3 * float(g["loss"])
    assert rel(eng._b["enc16"].float().cpu().view(g["enc"].shape), g["enc"]) < 1e-2
    # dlogits overwrote logits in the training pass; recompute forward-only for the logits check
    _, logits = eng.forward(xd, ld, train=False)
    lg = logits.float().cpu().numpy()
    assert rel(lg, g["logits"]) < 1e-2
    sure = g["top_margin"] > 0.05
    assert np.array_equal(lg.argmax(-1)[sure], g["top1_id"][sure])
    got = engine_grads(eng, dims, 0)
    for k in ("model.encoder.conv1.0.weight", "model.encoder.conv1.0.bias", "model.encoder.conv1.2.bias",
              "model.encoder.conv2.bias"):
        assert rel(got[k], g["grad." + k]) < 3e-2, (k, rel(got[k], g["grad." + k]))
    for k in ("model.encoder.conv1.2.weight", "model.encoder.conv2.weight"):
        assert rel(got[k][:48, :48], g["gradblock." + k]) < 3e-2, k
        assert abs(got[k].double().norm().item() - float(g["gradnorm." + k])) < 3e-2 * float(g["gradnorm." + k])


def test_base_shape_matches_reference_golden(dev):
    """whisper-base dims, 208 channels, T=6000 (BASELINE configs[0]/[1] shape at B=2)."""
    g = np.load(os.path.join(G, "train_base208.npz"))
    dims = WHISPER_BASE
    eng, sd, _ = make_engine(dims, dev, 0)
    x, labels = synth_batch(dims, int(g["B"]), int(g["seed_d"]))
    assert np.array_equal(labels, g["labels"])
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-3 * float(g["loss"]), (loss.item(), float(g["loss"]))
    enc = eng._b["enc16"].float().cpu().view(int(g["B"]), dims.src_pos, dims.d).numpy()
    assert rel(enc[:, ::97, :16], g["enc_slice"]) < 2e-2
    assert abs(np.sqrt((enc.astype(np.float64) ** 2).sum()) - float(g["enc_l2"])) < 1e-2 * float(g["enc_l2"])
    got = engine_grads(eng, dims, 0)
    for k in ("model.encoder.conv1.0.weight", "model.encoder.conv1.2.weight", "model.encoder.conv2.weight",
              "model.encoder.conv1.0.bias", "model.encoder.conv1.2.bias", "model.encoder.conv2.bias"):
        n = got[k].double().norm().item()
        assert abs(n - float(g["gradnorm." + k])) < 4e-2 * float(g["gradnorm." + k]), (k, n, float(g["gradnorm." + k]))
        assert rel(got[k].reshape(got[k].shape[0], -1)[:8, :8], g["gradslice." + k]) < 6e-2, k
    _, logits = eng.forward(xd, ld, train=False)
    lg = logits.float().cpu().numpy()
    assert rel(lg[:, :, :16], g["logits_slice"]) < 2e-2
    sure = g["top_margin"] > 0.1
    assert np.array_equal(lg.argmax(-1)[sure], g["top1_id"][sure])


def test_train_steps_reduce_loss_and_match_oracle_update(dev):
    """3 optimizer steps on one batch: loss falls; first update equals AdamW on the oracle's gradients."""
    from oracle import whisper_meg_oracle as O
    dims = TINY
    eng, sd, lora_sd = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    x, labels = synth_batch(dims, 4, 5)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    p0 = eng.P.clone()
    losses = [eng.train_step(xd, ld).item() for _ in range(4)]
    assert eng.step_dev.item() == 4 and eng.found_inf_dev.item() == 0
    assert losses[-1] < losses[0], losses
    # one fresh step: compare the parameter delta of conv2.bias with the oracle's AdamW (step 1: delta = -lr*sign(g) ~)
    eng2, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0, max_grad_norm=0.0)
    eng2.train_step(xd, ld)
    _, _, _, og = O.loss_and_grads(sd, lora_sd, x, labels, dims, 2.0)
    gref = og["model.encoder.conv2.bias"]
    pref, _, _ = O.adamw_reference(torch.from_numpy(sd["model.encoder.conv2.bias"]), gref, torch.zeros_like(gref),
                                   torch.zeros_like(gref), 1, 1e-3)
    got = eng2.pview("model.encoder.conv2.bias").cpu()
    big = gref.abs() > 1e-4 * gref.abs().max()   # step-1 AdamW is ~sign(g): ignore near-zero gradients
    torch.testing.assert_close(got[big], pref[big], atol=2e-5, rtol=0)


def test_adalora_forward_backward_vs_oracle(dev):
    """AdaLoRA at its initial rank 12 (finetune.py:206-208): y += B((A x) * E) * 32/(12+1e-5), loss += 0.5 * orth-reg;
    gradients of A, B, E and the conv stem against the oracle (rank padded to 16 inside the engine)."""
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    from oracle import whisper_meg_oracle as O
    dims, r = TINY, 12
    sd = make_state_dict(dims, 42)
    lora_sd = make_lora_state(dims, r, adalora=True, b_std=0.3)
    spec = LoraSpec(r=r, alpha=32.0, dropout=0.0, adalora=True, orth_reg_weight=0.5)
    eng = MegWhisperEngine(dims, sd, lora=spec, lora_sd=lora_sd, train_cfg=TrainCfg(), device=dev)
    x, labels = synth_batch(dims, 3, 11)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    o_loss, _, _, og = O.loss_and_grads(sd, lora_sd, x, labels, dims, spec.scale, orth_reg_weight=0.5)
    assert abs(loss.item() - o_loss.item()) <= 2e-3 * max(1.0, o_loss.item()), (loss.item(), o_loss.item())
    s = eng.loss_scale_dev.item()
    d, f, rp = dims.d, dims.ffn, eng.r
    bad = {}
    for i in range(dims.enc_layers):
        p = f"model.encoder.layers.{i}."
        A3 = eng.gview(p + "self_attn.qkv.lora_A").view(3, rp, d).cpu() / s
        E3 = eng.gview(p + "self_attn.qkv.lora_E").view(3, rp).cpu() / s
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            got = {"lora_A": A3[j, :r], "lora_B": eng.gview(p + f"self_attn.{nm}.lora_B").view(d, rp)[:, :r].cpu() / s,
                   "lora_E": E3[j, :r].reshape(r, 1)}
            for k, v in got.items():
                e = rel(v, og[p + f"self_attn.{nm}.{k}.weight"])
                if not e < 3e-2:
                    bad[p + nm + k] = e
            assert A3[j, r:].abs().max() == 0, "padded ranks must stay dead"
        for nm, no, ki in (("self_attn.out_proj", d, d), ("fc1", f, d), ("fc2", d, f)):
            got = {"lora_A": eng.gview(p + nm + ".lora_A").view(rp, ki)[:r].cpu() / s,
                   "lora_B": eng.gview(p + nm + ".lora_B").view(no, rp)[:, :r].cpu() / s,
                   "lora_E": eng.gview(p + nm + ".lora_E")[:r].reshape(r, 1).cpu() / s}
            for k, v in got.items():
                e = rel(v, og[p + nm + f".{k}.weight"])
                if not e < 3e-2:
                    bad[p + nm + k] = e
    assert not bad, bad
    assert rel(eng.conv_weight_grad("conv2").float().cpu() / s, og["model.encoder.conv2.weight"]) < 3e-2
    # and it trains: E starts at zero in the reference, so only E moves at first; here all three move
    l0 = eng.train_step(xd, ld).item()
    l1 = [eng.train_step(xd, ld).item() for _ in range(3)][-1]
    assert l1 < l0 and eng.found_inf_dev.item() == 0


def test_fine_tune_layers_adapts_only_the_first_layers(dev):
    """LoraSpec(layers=1) (finetune.py --fine_tune_layers=1): layer 0 carries adapters, layer 1 runs plain; loss and
    layer-0 adapter gradients against the oracle given the same (partial) adapter dict."""
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    from oracle import whisper_meg_oracle as O
    dims, r = TINY, 32
    sd = make_state_dict(dims, 42)
    lora_sd = {k: v for k, v in make_lora_state(dims, r, b_std=0.3).items() if ".layers.0." in k}
    eng = MegWhisperEngine(dims, sd, lora=LoraSpec(r=r, alpha=64.0, dropout=0.0, layers=1), lora_sd=lora_sd,
                           train_cfg=TrainCfg(), device=dev)
    assert not any(".layers.1." in n and "lora" in n for n in eng.seg_off)
    x, labels = synth_batch(dims, 3, 21)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    o_loss, _, _, og = O.loss_and_grads(sd, lora_sd, x, labels, dims, 2.0)
    assert abs(loss.item() - o_loss.item()) <= 2e-3 * max(1.0, o_loss.item())
    s = eng.loss_scale_dev.item()
    p = "model.encoder.layers.0."
    assert rel(eng.gview(p + "fc2.lora_B").view(dims.d, r).cpu() / s, og[p + "fc2.lora_B.weight"]) < 3e-2
    assert rel(eng.gview(p + "self_attn.out_proj.lora_A").view(r, dims.d).cpu() / s, og[p + "self_attn.out_proj.lora_A.weight"]) < 3e-2
    assert rel(eng.conv_weight_grad("conv2").float().cpu() / s, og["model.encoder.conv2.weight"]) < 3e-2


def test_gradient_accumulation_equals_one_big_batch(dev):
    """two micro-batches of 2 through accumulate_step == the gradients of the concatenated batch of 4 (dropout off,
    same number of label tokens per micro-batch so that the mean of means is the mean)."""
    dims = TINY
    eng, sd, lora_sd = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    x, labels = synth_batch(dims, 4, 5, min_k=12, max_k=12)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    g_big = eng.G.clone()
    p0 = eng.P.clone()
    l0 = eng.accumulate_step(xd[:2], ld[:2], 0, 2)
    g_half = eng.G.clone()
    assert eng.step_dev.item() == 0 and torch.equal(eng.P, p0)           # no optimizer step yet
    # second micro-batch: stop before the optimizer by replaying its pieces
    seed = eng.drop_seed
    eng.forward(xd[2:], ld[2:], train=True, compute_grad=True)
    eng.backward()
    g_acc = eng.G * 0.5
    assert rel(g_acc, g_big) < 2e-2, rel(g_acc, g_big)
    assert g_half.abs().sum() > 0 and l0.item() > 0
    # and the real thing steps the optimizer exactly once
    eng.zero_grad()
    eng.accumulate_step(xd[:2], ld[:2], 0, 2)
    eng.accumulate_step(xd[2:], ld[2:], 1, 2)
    assert eng.step_dev.item() == 1 and not torch.equal(eng.P, p0)


def test_replace_frontend_forward_backward_vs_oracle(dev):
    """projection_module('replace') (utils/model_utils.py:18-20): one Conv1d(ch, d, k3, s2) front-end."""
    from neuspeech1_amd.engine import MegWhisperEngine, TrainCfg
    from oracle import whisper_meg_oracle as O
    dims = TINY
    sd = make_state_dict(dims, 42, frontend="replace")
    eng = MegWhisperEngine(dims, sd, train_cfg=TrainCfg(), device=dev)
    assert eng.frontend == "replace"
    x, labels = synth_batch(dims, 3, 31)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    o_loss, o_logits, o_enc, og = O.loss_and_grads(sd, None, x, labels, dims, 0.0)
    assert abs(loss.item() - o_loss.item()) <= 2e-3 * max(1.0, o_loss.item()), (loss.item(), o_loss.item())
    assert rel(eng._b["enc16"].float().cpu().view(3, dims.src_pos, dims.d), o_enc) < 1e-2
    s = eng.loss_scale_dev.item()
    for nm in ("conv1", "conv2"):
        assert rel(eng.conv_weight_grad(nm).float().cpu() / s, og[f"model.encoder.{nm}.weight"]) < 3e-2, nm
        assert rel(eng.gview(f"model.encoder.{nm}.bias").cpu() / s, og[f"model.encoder.{nm}.bias"]) < 3e-2, nm
    l0 = eng.train_step(xd, ld).item()
    l1 = [eng.train_step(xd, ld).item() for _ in range(3)][-1]
    assert l1 < l0


def test_rccl_reducer_on_the_side_stream_single_rank(dev):
    """The DP plumbing on a real GPU: RCCL (backend "nccl") all-reduce AVG of the gradient chunks on the side stream,
    gated by events, with one rank (the collective is the identity): the steps must equal the plain steps up to the
    run-to-run last-bit noise of the fp32-atomic weight-gradient sums."""
    import os
    import socket
    import torch.distributed as dist
    from neuspeech1_amd.dp import GradReducer
    dims = TINY
    x, labels = synth_batch(dims, 3, 77)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng_a, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    eng_b, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        red = GradReducer(eng_b.G, force=True)
        calls = []
        def on_ready(lo, hi):
            calls.append((lo, hi))
            red.on_ready(lo, hi)
        for _ in range(3):
            # gradients (what the collective touches) are compared before the optimizer: AdamW's g / sqrt(v) turns the
            # last-bit noise of the fp32-atomic weight-gradient sums into O(lr) differences on near-zero entries
            eng_a.zero_grad()
            la, _ = eng_a.forward(xd, ld, train=True, compute_grad=True)
            eng_a.backward()
            eng_b.zero_grad()
            lb, _ = eng_b.forward(xd, ld, train=True, compute_grad=True)
            eng_b.backward(on_ready)
            red.finish()
            torch.cuda.synchronize()
            assert rel(eng_b.G, eng_a.G) < 1e-3
            assert abs(la.item() - lb.item()) < 2e-3 * max(1.0, abs(la.item()))
            eng_a.optimizer_step()
            eng_b.optimizer_step()
            # keep the replicas identical for the next comparison (see above: the optimizer amplifies last-bit noise)
            eng_b.P.copy_(eng_a.P); eng_b.M1.copy_(eng_a.M1); eng_b.M2.copy_(eng_a.M2)
            eng_b.refresh_operands()
        assert len(calls) == 9 and calls[0][0] < calls[0][1]          # 2 adapter chunks + the conv stem, per step
        assert sorted(calls[:3])[0][0] == 0 and max(h for _, h in calls[:3]) == eng_b.n_train
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("adalora", [False, True])
def test_full_model_adapters_vs_oracle(dev, adalora):
    """--ft_full (finetune.py:191-192): adapters on every decoder projection too (self q/k/v/out, cross q/k/v/out,
    fc1, fc2).  Loss, encoder AND decoder adapter gradients and the conv stem against the oracle; then it trains with
    dropout on."""
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg, _dec_sites
    from oracle import whisper_meg_oracle as O
    dims, r = TINY, (12 if adalora else 16)
    sd = make_state_dict(dims, 42)
    # moderate adapter size: with ten adapted sites per decoder layer a B of std 0.3 dominates the frozen weights and the
    # fp16 gradient error compounds to 10-20 % by the bottom layer (0.1-0.5 % at this size)
    lora_sd = make_lora_state(dims, r, adalora=adalora, b_std=0.1 if adalora else 0.05, decoder=True)
    spec = LoraSpec(r=r, alpha=32.0, dropout=0.0, adalora=adalora, orth_reg_weight=0.5 if adalora else 0.0, decoder=True)
    eng = MegWhisperEngine(dims, sd, lora=spec, lora_sd=lora_sd, train_cfg=TrainCfg(), device=dev)
    x, labels = synth_batch(dims, 3, 11)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    o_loss, _, _, og = O.loss_and_grads(sd, lora_sd, x, labels, dims, spec.scale, orth_reg_weight=spec.orth_reg_weight)
    assert abs(loss.item() - o_loss.item()) <= 2e-3 * max(1.0, o_loss.item()), (loss.item(), o_loss.item())
    s = eng.loss_scale_dev.item()
    d, f, rp = dims.d, dims.ffn, eng.r
    bad, n = {}, 0
    for i in range(dims.dec_layers):
        p = f"model.decoder.layers.{i}."
        for site, _, projs, kin, nout, _ in _dec_sites(d, f):
            G = len(projs)
            A = eng.gview(p + site + ".lora_A").view(G, rp, kin).cpu() / s
            E = eng.gview(p + site + ".lora_E").view(G, rp).cpu() / s if adalora else None
            for j, pj in enumerate(projs):
                got = {"lora_A": A[j, :r], "lora_B": eng.gview(p + pj + ".lora_B").view(nout, rp)[:, :r].cpu() / s}
                if adalora:
                    got["lora_E"] = E[j, :r].reshape(r, 1)
                for k, v in got.items():
                    e = rel(v, og[p + pj + f".{k}.weight"])
                    n += 1
                    if not e < 3e-2:
                        bad[p + pj + "." + k] = e
                assert rp == r or A[j, r:].abs().max() == 0, "padded ranks must stay dead"
    assert n == dims.dec_layers * 10 * (3 if adalora else 2)
    # the encoder side still matches with the decoder adapters in the graph (its input gradient now carries the cross
    # K/V adapters' contribution)
    for i in range(dims.enc_layers):
        p = f"model.encoder.layers.{i}."
        for nm, no, ki in (("self_attn.out_proj", d, d), ("fc1", f, d), ("fc2", d, f)):
            for k, v in (("lora_A", eng.gview(p + nm + ".lora_A").view(rp, ki)[:r].cpu() / s),
                         ("lora_B", eng.gview(p + nm + ".lora_B").view(no, rp)[:, :r].cpu() / s)):
                e = rel(v, og[p + nm + f".{k}.weight"])
                if not e < 3e-2:
                    bad[p + nm + "." + k] = e
    assert not bad, bad
    assert rel(eng.conv_weight_grad("conv2").float().cpu() / s, og["model.encoder.conv2.weight"]) < 3e-2
    # trains with dropout on; the eval forward through the adapters agrees with the oracle's logits
    torch.manual_seed(0)
    eng2 = MegWhisperEngine(dims, sd, lora=LoraSpec(r=r, alpha=32.0, dropout=0.1, adalora=adalora, decoder=True), lora_sd=lora_sd,
                            train_cfg=TrainCfg(lr=1e-3, warmup_steps=0, total_steps=0), device=dev)
    _, lg = eng2.forward(xd, ld, train=False)
    _, o_logits, _ = O.forward({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, torch.from_numpy(x), dims,
                               labels=torch.from_numpy(labels), lora={k: torch.from_numpy(v) for k, v in lora_sd.items()},
                               scale=spec.scale)
    assert rel(lg.float().cpu(), o_logits) < 1e-2
    ls = [eng2.train_step(xd, ld).item() for _ in range(5)]
    assert ls[-1] < ls[0] and eng2.found_inf_dev.item() == 0, ls


def test_base_shape_lora_forward_backward_vs_oracle(dev):
    """LoRA r = 32 at the BASELINE dims (whisper-base, 208 channels, T = 6000; B = 1 so the CPU oracle stays in seconds):
    the phase-interleaved / ring GEMM kernels and the full-size attention kernels, not the small-shape fallbacks, against
    the oracle's loss and adapter gradients."""
    from oracle import whisper_meg_oracle as O
    dims = WHISPER_BASE
    eng, sd, lora_sd = make_engine(dims, dev, 32)
    x, labels = synth_batch(dims, 1, 31)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    o_loss, _, _, og = O.loss_and_grads(sd, lora_sd, x, labels, dims, 2.0)
    assert abs(loss.item() - o_loss.item()) <= 2e-3 * max(1.0, o_loss.item()), (loss.item(), o_loss.item())
    got = engine_grads(eng, dims, 32)
    bad = {}
    for k, ref in og.items():
        if k in got and ("lora" in k or "bias" in k):
            e = rel(got[k], ref)
            if not e < 4e-2:
                bad[k] = e
    assert not bad, bad
    assert len([k for k in og if "lora" in k]) == dims.enc_layers * 12


def test_base_shape_step_is_the_same_through_the_big_gemm_kernels(dev):
    """At the bench size the encoder GEMMs dispatch to the phase-interleaved 256^2 kernel (M >= 2048 and >= 192 tiles),
    which the B = 1 / 2 oracle cases never reach: one LoRA step with dropout at whisper-base dims, B = 8, under the
    automatic dispatch against the same step forced through the register-staged kernel (the one the oracle cases pin)."""
    from neuspeech1_amd import lib
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    dims = WHISPER_BASE
    sd = make_state_dict(dims, 42)
    lora_sd = make_lora_state(dims, 32)
    x, labels = synth_batch(dims, 8, 55)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    res = {}
    try:
        for mode in (0, 1):
            lib.load().ns_debug_set_ring(mode)
            eng = MegWhisperEngine(dims, sd, lora=LoraSpec(r=32, alpha=64.0, dropout=0.05), lora_sd=lora_sd,
                                   train_cfg=TrainCfg(lr=1e-3), device=dev)
            eng.drop_seed = 777
            eng.no_side_u2 = mode == 0       # mode 1 also takes fc2's adapter bottleneck from fc1's GELU epilogue (side product)
            eng.zero_grad()
            loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
            eng.backward()
            assert ("u2_slabs" in eng._b) == (mode == 1)
            res[mode] = (loss.item(), eng.G.clone(), eng._b["enc16"].float().clone())
            del eng
    finally:
        lib.load().ns_debug_set_ring(1)
    assert abs(res[0][0] - res[1][0]) < 1e-3 * max(1.0, abs(res[0][0])), (res[0][0], res[1][0])
    assert rel(res[1][2], res[0][2]) < 2e-3
    assert rel(res[1][1], res[0][1]) < 1e-2


def test_adalora_matches_reference_on_merged_weights(dev):
    """The reference's DEFAULT adapter (finetune.py:43,205-208: AdaLoRA, init_r 12, alpha 32) against the reference
    object itself: the engine's adapter path (rank padded to 16, diag(E) folded into the fp16 up-projection) must give
    the logits stock HF gives on W + alpha/(r+1e-5) B (A * E) (tests/golden/adalora_merged_tiny.npz), and
    merge_and_unload's arithmetic (PeftModel) the same again."""
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    g = np.load(os.path.join(G, "adalora_merged_tiny.npz"))
    dims, r = TINY, int(g["r"])
    sd = make_state_dict(dims, 42)
    lora_sd = make_lora_state(dims, r, adalora=True, b_std=float(g["b_std"]))
    spec = LoraSpec(r=r, alpha=float(g["alpha"]), dropout=0.1, adalora=True, orth_reg_weight=0.5)
    eng = MegWhisperEngine(dims, sd, lora=spec, lora_sd=lora_sd, train_cfg=TrainCfg(), device=dev)
    x, labels = synth_batch(dims, int(g["B"]), 1234)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    loss, logits = eng.forward(xd, ld, train=False)           # eval mode: dropout off, no regulariser in the loss
    assert abs(loss.item() - float(g["loss"])) < 2e-3 * float(g["loss"]), (loss.item(), float(g["loss"]))
    assert abs(loss.item() - float(g["loss_base"])) > 1e-2
    lg = logits.float().cpu().numpy()[:, :, ::3]
    assert rel(lg, g["logits"]) < 1e-2, rel(lg, g["logits"])
    # the frozen model through the same engine class is NOT this function
    eng0 = MegWhisperEngine(dims, sd, train_cfg=TrainCfg(), device=dev)
    loss0, _ = eng0.forward(xd, ld, train=False)
    assert abs(loss0.item() - float(g["loss_base"])) < 2e-3 * float(g["loss_base"])


@pytest.mark.parametrize("tag", ["base273", "lv2w"])
def test_273_channels_and_large_v2_width_match_reference_golden(dev, tag):
    """BASELINE configs[3] shape (whisper-base, 273 channels: ch_pad 288, first conv K = 864; /root/reference
    README.md:60-64) and configs[4]'s WIDTH (d 1280, 20 heads, ffn 5120; 2 + 2 layers), B = 1: loss, encoder states,
    logits and the conv-stem gradients against the reference object (tests/golden/train_{base273,lv2w}.npz)."""
    from neuspeech1_amd.weights import LV2W, WhisperDims
    g = np.load(os.path.join(G, f"train_{tag}.npz"))
    dims = LV2W if tag == "lv2w" else WhisperDims(ch=273)
    eng, sd, _ = make_engine(dims, dev, 0)
    x, labels = synth_batch(dims, int(g["B"]), int(g["seed_d"]))
    assert np.array_equal(labels, g["labels"])
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-3 * float(g["loss"]), (loss.item(), float(g["loss"]))
    enc = eng._b["enc16"].float().cpu().view(int(g["B"]), dims.src_pos, dims.d).numpy()
    assert rel(enc[:, ::97, :16], g["enc_slice"]) < 2e-2
    assert abs(np.sqrt((enc.astype(np.float64) ** 2).sum()) - float(g["enc_l2"])) < 1e-2 * float(g["enc_l2"])
    got = engine_grads(eng, dims, 0)
    for k in ("model.encoder.conv1.0.weight", "model.encoder.conv1.2.weight", "model.encoder.conv2.weight",
              "model.encoder.conv1.0.bias", "model.encoder.conv1.2.bias", "model.encoder.conv2.bias"):
        n = got[k].double().norm().item()
        assert abs(n - float(g["gradnorm." + k])) < 4e-2 * float(g["gradnorm." + k]), (k, n, float(g["gradnorm." + k]))
        assert rel(got[k].reshape(got[k].shape[0], -1)[:8, :8], g["gradslice." + k]) < 6e-2, k
    _, logits = eng.forward(xd, ld, train=False)
    lg = logits.float().cpu().numpy()
    assert rel(lg[:, :, :16], g["logits_slice"]) < 2e-2
    sure = g["top_margin"] > 0.1
    assert np.array_equal(lg.argmax(-1)[sure], g["top1_id"][sure])


def test_large_v2_width_lora_forward_backward_vs_oracle(dev):
    """LoRA r = 32 at whisper-large-v2's width (d 1280: five 256-column tiles, 20 heads, ffn 5120; 2 + 2 layers, 273-ch,
    B = 1): loss and every adapter gradient against the oracle."""
    from neuspeech1_amd.weights import LV2W
    from oracle import whisper_meg_oracle as O
    dims = LV2W
    eng, sd, lora_sd = make_engine(dims, dev, 32)
    x, labels = synth_batch(dims, 1, 31)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    o_loss, _, _, og = O.loss_and_grads(sd, lora_sd, x, labels, dims, 2.0)
    assert abs(loss.item() - o_loss.item()) <= 2e-3 * max(1.0, o_loss.item()), (loss.item(), o_loss.item())
    got = engine_grads(eng, dims, 32)
    bad = {k: rel(got[k], ref) for k, ref in og.items() if k in got and ("lora" in k or "bias" in k) and not rel(got[k], ref) < 4e-2}
    assert not bad, bad


# ------------------------------------------------------------------ BASELINE configs[4] at FULL depth
@pytest.fixture(scope="module")
def lv2_state():
    from neuspeech1_amd.weights import WHISPER_LARGE_V2
    return make_state_dict(WHISPER_LARGE_V2, 42)          # 1.54 G parameters from the counter-based generator


def test_large_v2_full_depth_matches_reference_golden(dev, lv2_state):
    """whisper-large-v2 as BASELINE configs[4] names it -- 32 + 32 layers, d 1280, 20 heads, ffn 5120, 273 channels -- B = 1:
    loss, encoder states, logits and the conv-stem gradients against the reference object run at full depth in the build
    container (tests/golden/train_lv2.npz, tools/make_goldens.py lv2; reference recipes README.md:67-125).  Tolerances are
    ~4x what was measured (loss 3e-5, encoder slice 7.9e-4, gradient norms 2.8e-4, gradient blocks 1.4e-3 relative): 64
    layers of fp16 re-rounding average out rather than add up."""
    from neuspeech1_amd.engine import MegWhisperEngine, TrainCfg
    from neuspeech1_amd.weights import WHISPER_LARGE_V2
    g = np.load(os.path.join(G, "train_lv2.npz"))
    dims = WHISPER_LARGE_V2
    eng = MegWhisperEngine(dims, lv2_state, train_cfg=TrainCfg(), device=dev)
    x, labels = synth_batch(dims, int(g["B"]), int(g["seed_d"]))
    assert np.array_equal(labels, g["labels"])
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    assert abs(loss.item() - float(g["loss"])) < 3e-4 * float(g["loss"]), (loss.item(), float(g["loss"]))
    enc = eng._b["enc16"].float().cpu().view(int(g["B"]), dims.src_pos, dims.d).numpy()
    e_sl = rel(enc[:, ::97, :16], g["enc_slice"])
    assert e_sl < 3e-3, e_sl
    assert abs(np.sqrt((enc.astype(np.float64) ** 2).sum()) - float(g["enc_l2"])) < 2e-3 * float(g["enc_l2"])
    got = engine_grads(eng, dims, 0)
    worst = {}
    for k in ("model.encoder.conv1.0.weight", "model.encoder.conv1.2.weight", "model.encoder.conv2.weight",
              "model.encoder.conv1.0.bias", "model.encoder.conv1.2.bias", "model.encoder.conv2.bias"):
        n = got[k].double().norm().item()
        worst[k] = (abs(n - float(g["gradnorm." + k])) / float(g["gradnorm." + k]),
                    rel(got[k].reshape(got[k].shape[0], -1)[:8, :8], g["gradslice." + k]))
    print(f"\nlarge-v2 full depth: loss {loss.item():.5f} vs {float(g['loss']):.5f}, enc slice rel {e_sl:.2e}, conv grads {worst}")
    assert all(a < 2e-3 and b < 6e-3 for a, b in worst.values()), worst
    del eng._bufs
    eng._bufs = {}
    _, logits = eng.forward(xd, ld, train=False)
    lg = logits.float().cpu().numpy()
    assert rel(lg[:, :, :16], g["logits_slice"]) < 5e-3
    sure = g["top_margin"] > 0.05
    assert np.array_equal(lg.argmax(-1)[sure], g["top1_id"][sure])


def test_large_v2_full_depth_lora_gradients_match_oracle_golden(dev, lv2_state):
    """LoRA r = 32 on all 192 encoder projections of the full-depth model: loss and every adapter / conv gradient against
    the oracle's (tests/golden/lora_oracle_lv2.npz: per-tensor norms + leading 8 x 8 blocks; the oracle takes minutes at
    this size, so it ran in the build container).  The oracle's adapter arithmetic is pinned on the reference object by
    the merged-weight goldens."""
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    from neuspeech1_amd.weights import WHISPER_LARGE_V2
    g = np.load(os.path.join(G, "lora_oracle_lv2.npz"))
    dims, r = WHISPER_LARGE_V2, int(g["r"])
    eng = MegWhisperEngine(dims, lv2_state, lora=LoraSpec(r=r, alpha=float(g["alpha"]), dropout=0.0),
                           lora_sd=make_lora_state(dims, r), train_cfg=TrainCfg(), device=dev)
    x, labels = synth_batch(dims, int(g["B"]), int(g["seed_d"]))
    assert np.array_equal(labels, g["labels"])
    eng.zero_grad()
    loss, _ = eng.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), train=True, compute_grad=True)
    eng.backward()
    assert abs(loss.item() - float(g["loss"])) < 3e-4 * float(g["loss"]), (loss.item(), float(g["loss"]))
    got = engine_grads(eng, dims, r)
    names = [str(n) for n in g["names"]]
    assert len(names) == 2 * 6 * dims.enc_layers + 6
    bad, wn, wb = {}, 0.0, 0.0
    for i, k in enumerate(names):
        t = got[k]
        en = abs(t.double().norm().item() - float(g["gradnorm"][i])) / float(g["gradnorm"][i])
        t2 = t.reshape(t.shape[0], -1)
        blk = g["gradblock"][i][:min(8, t2.shape[0]), :min(8, t2.shape[1])]
        # the block's error against the tensor's typical magnitude (an 8 x 8 block of near-zero entries has no relative scale)
        rms = float(g["gradnorm"][i]) / np.sqrt(t.numel())
        eb = float(np.linalg.norm(t2[:8, :8].double().numpy() - blk)) / (rms * np.sqrt(blk.size))
        wn, wb = max(wn, en), max(wb, eb)
        # (measured: norms within 4.6e-3; blocks within 2e-2 except the q / k adapters of the top layers, 3.6e-2 .. 5.8e-2 of the
        # tensor's RMS: 64 entries of a gradient that passed 30 softmax backward stages in fp16)
        if not (en < 1.5e-2 and eb < 8e-2):
            bad[k] = (en, eb)
    print(f"\nlarge-v2 full depth LoRA: loss {loss.item():.5f} vs {float(g['loss']):.5f}; worst norm err {wn:.2e}, worst block err {wb:.2e}")
    assert not bad, bad


def test_graph_replayed_train_step_equals_eager(dev):
    """engine.train_step replays a captured hipGraph from the third step of a shape on.  Same seeds, same data: every step's
    loss and gradient buffer must equal the eager engine's (up to the run-to-run last-bit noise of the fp32-atomic weight
    gradient sums, like two eager runs), with LoRA dropout ON -- the masks come from the device-resident step counter
    (ns_gemm_desc.seed_dev), so a replayed step must draw the masks of ITS step, not of the captured one."""
    dims = TINY
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    sd, lora_sd = make_state_dict(dims, 42), make_lora_state(dims, 32)
    x, labels = synth_batch(dims, 3, 77)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    mk = lambda: MegWhisperEngine(dims, sd, lora=LoraSpec(r=32, alpha=64.0, dropout=0.05), lora_sd=lora_sd,  # noqa: E731
                                  train_cfg=TrainCfg(lr=1e-3, warmup_steps=2, total_steps=50), device=dev)
    a, b = mk(), mk()
    a.use_graph = False
    assert b.use_graph
    g_prev = None
    for step in range(6):
        la = a.train_step(xd, ld).item()
        lb = b.train_step(xd, ld).item()
        assert abs(la - lb) < 1e-3 * max(1.0, abs(la)), (step, la, lb)
        assert rel(b.G, a.G) < 1e-3, (step, rel(b.G, a.G))
        assert a.seed_ctr.item() == b.seed_ctr.item() == step + 1 and a.step_dev.item() == b.step_dev.item()
        if g_prev is not None:      # the gradients do move from step to step (new masks, new weights)
            assert rel(b.G, g_prev) > 1e-2
        g_prev = b.G.clone()
        # keep the replicas identical: AdamW's g / sqrt(v) turns last-bit gradient noise into O(lr) differences
        b.P.copy_(a.P); b.M1.copy_(a.M1); b.M2.copy_(a.M2)
        b.refresh_operands()
    assert len(b._graphs) == 1 and len(a._graphs) == 0
    # the masks of a step are a function of the device counter alone: same counter, same weights -> identical forward
    def u_of(ctr):
        b.seed_ctr.fill_(ctr)
        b.forward(xd, ld, train=True, compute_grad=False)
        return b._b["uqkv"][0].clone()
    u5, u5b, u6 = u_of(5), u_of(5), u_of(6)
    assert torch.equal(u5, u5b) and not torch.equal(u5, u6)


def test_graph_capture_with_gradient_exchange_hooks(dev):
    """With a gradient exchange the captured step is cut at backward's on_ready points: the hooks must fire in order with
    the eager step's (lo, hi) ranges, between the replays, and reduce_fn before the optimizer segment."""
    dims = TINY
    eng, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    x, labels = synth_batch(dims, 3, 77)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    log = []
    on_ready = lambda lo, hi: log.append((lo, hi))  # noqa: E731
    reduce_fn = lambda: log.append("reduce")        # noqa: E731
    eng.use_graph = False
    eng.train_step(xd, ld, on_ready=on_ready, reduce_fn=reduce_fn)
    eager_log, log = log, []
    eng.use_graph = True
    for _ in range(3):
        log.clear()
        p_before = eng.P.clone()
        eng.train_step(xd, ld, on_ready=on_ready, reduce_fn=reduce_fn)
        assert log == eager_log, (log, eager_log)
        assert not torch.equal(eng.P, p_before)
    assert len(eng._graphs) == 1 and len(next(iter(eng._graphs.values()))["segs"]) == len(eager_log) + 1


def test_graph_replay_with_rccl_exchange_single_rank(dev):
    """What runs on an 8-GPU node, minus the wire: an RCCL process group alive (its watchdog thread included), the step
    replayed from hipGraphs CUT at the exchange points, dp.GradReducer launching the chunks' all-reduce (AVG) eagerly on its
    side stream between the replays and the optimizer segment waiting for them.  World size 1 (the collective is the
    identity), so the replayed steps must equal an eager engine's; the capture must succeed once and never fail
    (finetune.py:115-122,248 -> DDP in the reference)."""
    import os
    import socket
    import torch.distributed as dist
    from neuspeech1_amd.dp import GradReducer
    dims = TINY
    x, labels = synth_batch(dims, 3, 77)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng_a, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    eng_b, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    eng_a.use_graph = False
    assert eng_b.use_graph
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        red = GradReducer(eng_b.G, force=True, timing=True)
        dist.all_reduce(torch.ones(4, device=dev))      # communicator (and its watchdog) up before the first capture
        torch.cuda.synchronize()
        for step in range(9):
            la = eng_a.train_step(xd, ld).item()
            lb = eng_b.train_step(xd, ld, on_ready=red.on_ready, reduce_fn=red.finish).item()
            assert abs(la - lb) < 1e-3 * max(1.0, abs(la)), (step, la, lb)
            assert rel(eng_b.G, eng_a.G) < 1e-3, (step, rel(eng_b.G, eng_a.G))
            eng_b.P.copy_(eng_a.P); eng_b.M1.copy_(eng_a.M1); eng_b.M2.copy_(eng_a.M2)
            eng_b.refresh_operands()
        torch.cuda.synchronize()
        st = eng_b.graph_stats()
        assert st["enabled"] and st["graphs_cached"] == 1 and st["captures"] == 1 and st["capture_failures"] == 0, st
        assert st["replays"] == 8, st                      # first step of a shape eager, then capture + replay
        # 3 exchange cuts (two adapter chunks + the conv stem) + "reduce" + the optimizer = 5 segments
        assert st["segments"] == [5], st
        assert red.total_ms() > 0.0 and red.exposed_ms()[1] == 4 * eng_b.n_train
    finally:
        dist.destroy_process_group()


def test_failed_capture_falls_back_and_restores_the_stream(dev):
    """torch.cuda.graph.__exit__ ends the capture BEFORE it leaves its side stream, so a capture_end that raises (a capture
    invalidated by a foreign host thread) leaves the capture stream current.  train_step must put the caller's stream back,
    run the step eagerly, count and report the failure, and stop using graphs for the run.  The failure is SIMULATED at
    _capture_step's boundary (the side stream left current + the RuntimeError torch raises): really invalidating a capture
    makes this torch / HIP stack abort later, inside the allocator, when the tensors of the dead capture are released."""
    dims = TINY
    eng, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    x, labels = synth_batch(dims, 3, 77)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.train_step(xd, ld)                       # warm step (eager)
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream(dev)
    real = eng._capture_step

    def failing_capture(*a, **k):
        torch.cuda.set_stream(side)              # what the un-exited stream context of torch.cuda.graph leaves behind
        raise RuntimeError("HIP error: operation failed due to a previous error during capture")
    eng._capture_step = failing_capture
    with pytest.warns(UserWarning, match="capture failed"):
        l1 = eng.train_step(xd, ld)
    assert torch.cuda.current_stream() == cur
    assert torch.isfinite(l1).all()
    # ONE real failure ends graph use for the run (recovery from a truly invalidated capture cannot be relied on, and
    # under DP a rank that aborts hangs the others): the step still runs, eagerly, and the stats say so
    st = eng.graph_stats()
    assert st["capture_failures"] == 1 and st["captures"] == 0 and not st["enabled"]
    eng._capture_step = real
    for _ in range(3):
        l2 = eng.train_step(xd, ld)
    assert torch.cuda.current_stream() == cur and torch.isfinite(l2).all()
    st = eng.graph_stats()
    assert st["captures"] == 0 and st["replays"] == 0 and st["capture_failures"] == 1 and not st["enabled"], st


def test_graph_replay_path_checks_the_batch_like_the_eager_path(dev):
    """only the first batch of a shape goes through encode()'s asserts; a later fp16 / non-contiguous / CPU batch of the
    same shape would be packed from garbage by the eager ns_signal_pack in front of the replay.  It must raise instead."""
    dims = TINY
    eng, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    x, labels = synth_batch(dims, 3, 77)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    for _ in range(3):
        eng.train_step(xd, ld)
    assert eng.graph_stats()["replays"] >= 1
    with pytest.raises(ValueError, match="contiguous float32"):
        eng.train_step(xd.half(), ld)
    with pytest.raises(ValueError, match="contiguous float32"):
        eng.train_step(xd.transpose(1, 2).contiguous().transpose(1, 2), ld)
    with pytest.raises(ValueError, match="contiguous float32"):
        eng.train_step(xd.cpu(), ld)
    assert torch.isfinite(eng.train_step(xd, ld)).all()


def test_plain_tensor_batches_share_one_graph(dev):
    """Plain fp32 batches arrive as a fresh tensor (a new address) every step: the graph path packs them eagerly into the
    engine's static xin, so ONE capture serves them all, and reduce_fn alone (no on_ready) still gets its hook."""
    dims = TINY
    eng, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    ref, _, _ = make_engine(dims, dev, 32, lr=1e-3, warmup_steps=0, total_steps=0)
    ref.use_graph = False
    calls = []
    _, labels = synth_batch(dims, 3, 100)        # one label shape: the signal is what arrives in a fresh tensor every step
    ld = torch.from_numpy(labels).to(dev)
    for s in range(6):
        x, _ = synth_batch(dims, 3, 100 + s)
        xd = torch.from_numpy(x).to(dev).clone()
        lg = eng.train_step(xd, ld, reduce_fn=lambda: calls.append(s)).item()
        le = ref.train_step(xd, ld).item()
        assert abs(lg - le) < 1e-3 * max(1.0, abs(le)), (s, lg, le)
        assert rel(eng.G, ref.G) < 1e-3
        eng.P.copy_(ref.P); eng.M1.copy_(ref.M1); eng.M2.copy_(ref.M2)
        eng.refresh_operands()
    st = eng.graph_stats()
    assert st["graphs_cached"] == 1 and st["captures"] == 1 and st["replays"] == 5, st
    assert calls == list(range(6))               # eager warm step + every replay


def test_adalora_graph_replays_back_to_back_report_a_finite_regulariser(dev):
    """The reference's default adapter under graph replay WITHOUT host synchronisation between steps (how finetune.py and
    bench.py run): every step's reported loss (cross-entropy + orthogonality regulariser) must stay finite and follow the
    eager engine's.  Regression: ns_orth_reg cleared its Gram workspace with hipMemsetAsync; captured in a hipGraph that
    memset node was not reliably ordered against the kernels around it under back-to-back replays (a garbage / inf
    regulariser every few dozen steps at whisper-base size, never with eager launches)."""
    from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
    dims = WHISPER_BASE
    sd = make_state_dict(dims, 42)
    mk = lambda: MegWhisperEngine(dims, sd, lora=LoraSpec(r=12, alpha=32.0, dropout=0.1, adalora=True),  # noqa: E731
                                  train_cfg=TrainCfg(lr=1e-3, warmup_steps=20, total_steps=200), device=dev)
    pool = []
    for i in range(3):
        x, labels = synth_batch(dims, 8, 1000 + i)
        pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)))
    out = {}
    for graph in (False, True):
        torch.manual_seed(42)
        eng = mk()
        eng.use_graph = graph
        hist = []
        for s in range(60):
            loss = eng.train_step(*pool[s % 3])
            hist.append(loss.clone())            # a device copy: no host synchronisation inside the loop
        torch.cuda.synchronize()
        out[graph] = torch.cat([h.reshape(1) for h in hist]).cpu()
        assert graph == (len(eng._graphs) > 0)
    assert torch.isfinite(out[True]).all(), out[True]
    # same seeds, same data, same dropout counter: the two trajectories differ only by the fp32-atomic noise that AdamW
    # amplifies over 60 steps
    assert (out[True] - out[False]).abs().max() < 5e-2 * out[False].abs().max(), (out[True], out[False])
    assert out[True][-1] < out[True][0]
