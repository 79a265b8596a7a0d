"""__graft_entry__.smoke(): one tiny forward+backward+optimizer step and one short greedy / beam decode of the hot path
on cuda:0 through the HIP library, checked against the CPU oracle (the oracle is the checker only, never the product
path)."""
import torch


def run_smoke():
    from .engine import LoraSpec, MegWhisperEngine, TrainCfg
    from .weights import TINY, make_lora_state, make_state_dict, synth_batch
    from . import lib
    lib.load()
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = torch.device("cuda:0")
    dims = TINY
    sd = make_state_dict(dims, 42)
    lora_sd = make_lora_state(dims, 32)
    eng = MegWhisperEngine(dims, sd, lora=LoraSpec(32, 64.0, 0.0), lora_sd=lora_sd, train_cfg=TrainCfg(lr=1e-3), device=dev)
    x, labels = synth_batch(dims, 2, 3)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    l0 = eng.train_step(xd, ld).item()
    l1 = eng.train_step(xd, ld).item()
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import whisper_meg_oracle as O
    ref, _, _, _ = O.loss_and_grads(sd, lora_sd, x, labels, dims, 2.0)
    assert abs(l0 - ref.item()) < 2e-3 * max(1.0, ref.item()), (l0, ref.item())
    assert eng.found_inf_dev.item() == 0 and l1 < l0, (l0, l1)
    print(f"[smoke] tiny train step ok: loss {l0:.4f} (oracle {ref.item():.4f}) -> {l1:.4f}")
    # decode: token ids of the greedy and the beam-search loop against the oracle's (frozen weights, no adapters)
    from .generate import Generator
    gen = Generator(MegWhisperEngine(dims, sd, device=dev))
    prompt = torch.from_numpy(labels[:, :4].copy())
    kw = dict(repetition_penalty=5.0, no_repeat_ngram_size=2)
    sdt = O.to_torch(sd)
    got = gen.generate(xd, prompt.to(dev), num_beams=1, max_new_tokens=8, check_every=1, **kw).cpu()
    want = O.greedy(sdt, torch.from_numpy(x), dims, prompt, 8, **kw)
    assert torch.equal(got, want), (got.tolist(), want.tolist())
    got = gen.generate(xd, prompt.to(dev), num_beams=3, max_new_tokens=6, check_every=1, **kw).cpu()
    want = O.beam_search(sdt, torch.from_numpy(x), dims, prompt, 3, 6, **kw)
    want = want[0] if isinstance(want, tuple) else want
    assert torch.equal(got[:, :want.shape[1]], want), (got.tolist(), want.tolist())
    print(f"[smoke] tiny greedy / beam-3 decode ok: ids {got[0, 4:].tolist()}")
