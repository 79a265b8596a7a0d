import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from neuspeech1_amd.engine import LoraSpec, MegWhisperEngine, TrainCfg
from neuspeech1_amd.generate import Generator
from neuspeech1_amd.weights import TINY, make_lora_state, make_state_dict, synth_batch
from oracle import whisper_meg_oracle as O
dev = torch.device("cuda:0")
dims = TINY
sd = make_state_dict(dims, 42); lora_sd = make_lora_state(dims, 16)
eng = MegWhisperEngine(dims, sd, lora=LoraSpec(16, 32.0, 0.0), lora_sd=lora_sd, train_cfg=TrainCfg(lr=1e-3), device=dev)
rng = np.random.default_rng(0)
for B, L in ((1, 5), (3, 1), (7, 2), (5, dims.tgt_pos), (2, 33)):
    x, _ = synth_batch(dims, B, 100 + B)
    labels = rng.integers(5, dims.vocab - 10, (B, L)).astype(np.int64)
    if L > 3:
        labels[0, L // 2:] = -100          # ragged: padded tail
    if B > 2:
        labels[1, :] = -100                # a row with no valid label at all
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    eng.zero_grad()
    loss, _ = eng.forward(xd, ld, train=True, compute_grad=True)
    eng.backward()
    ref, _, _, og = O.loss_and_grads(sd, lora_sd, x, labels, dims, 2.0)
    g = eng.gview("model.encoder.layers.0.fc1.lora_A").view(eng.r, dims.d)[:16].cpu() / eng.loss_scale_dev.item()
    rel = ((g - og["model.encoder.layers.0.fc1.lora_A.weight"]).norm() / og["model.encoder.layers.0.fc1.lora_A.weight"].norm()).item()
    print(f"B={B} L={L}: loss {loss.item():.5f} oracle {ref.item():.5f}  grad rel {rel:.4f}  finite {bool(torch.isfinite(eng.G).all())}")
gen = Generator(MegWhisperEngine(dims, sd, device=dev))
for B, P, nb, new in ((1, 1, 1, 5), (1, 4, 5, 3), (3, 2, 2, 40), (2, 4, 8, 59)):
    x, labels = synth_batch(dims, B, 7)
    out = gen.generate(torch.from_numpy(x).to(dev), torch.from_numpy(labels[:, :P].copy()).to(dev), num_beams=nb, max_new_tokens=new,
                       repetition_penalty=5.0, no_repeat_ngram_size=2, check_every=3)
    want = (O.greedy if nb == 1 else None)
    if nb == 1:
        ref = O.greedy(O.to_torch(sd), torch.from_numpy(x), dims, torch.from_numpy(labels[:, :P].copy()), new, repetition_penalty=5.0, no_repeat_ngram_size=2)
    else:
        ref = O.beam_search(O.to_torch(sd), torch.from_numpy(x), dims, torch.from_numpy(labels[:, :P].copy()), nb, new, repetition_penalty=5.0, no_repeat_ngram_size=2)
    o = out.cpu()
    n = min(o.shape[1], ref.shape[1])
    print(f"decode B={B} P={P} beams={nb} new={new}: shape {tuple(o.shape)} oracle {tuple(ref.shape)} ids equal {bool(torch.equal(o[:, :n], ref[:, :n]))}")
    if not torch.equal(o[:, :n], ref[:, :n]):
        for b in range(B):
            d = (o[b, :n] != ref[b, :n]).nonzero()
            print("   row", b, "first mismatch at", int(d[0]) if len(d) else None, "of", n, "| mine", o[b, max(0, int(d[0]) - 2):int(d[0]) + 3].tolist() if len(d) else "", "| oracle", ref[b, max(0, int(d[0]) - 2):int(d[0]) + 3].tolist() if len(d) else "")
        for nb2 in (6, 7, 8):
            o2 = gen.generate(torch.from_numpy(x).to(dev), torch.from_numpy(labels[:, :P].copy()).to(dev), num_beams=nb2, max_new_tokens=20, repetition_penalty=5.0, no_repeat_ngram_size=2, check_every=3).cpu()
            r2 = O.beam_search(O.to_torch(sd), torch.from_numpy(x), dims, torch.from_numpy(labels[:, :P].copy()), nb2, 20, repetition_penalty=5.0, no_repeat_ngram_size=2)
            print("   beams", nb2, "new 20: equal", bool(torch.equal(o2[:, :r2.shape[1]], r2)))
