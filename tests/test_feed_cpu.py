"""Host half of the on-GPU data feed (neuspeech1_amd/feed.py): header parsing, read planning, staging bytes and the
ns_feed_item table, checked by replaying the kernel's rules in numpy against the reader + collator path -- which
tests/test_dropin_cpu.py pins to the reference's own reader outputs (tests/golden/reader.npz)."""
import numpy as np
import pytest

from neuspeech1_amd.feed import ITEM_BYTES, RawSignal, _fill, item_table, layout_batch, npy_layout, plan_read
from neuspeech1_amd.synthetic import SyntheticProcessor
from neuspeech1_amd.weights import WHISPER_BASE
from tests.feed_cases import datasets, write_cases
from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding

NP_DT = {0: np.float64, 1: np.float32, 2: np.float16}


def replay(table, staging, base, ch, T):
    """what ns_feed_pack computes (include/neuspeech_hip.h), in numpy: (B, ch, T) float32"""
    out = np.zeros((len(table), ch, T), np.float32)
    for b, it in enumerate(table):
        rows, n, ld = min(int(it["rows"]), ch), min(int(it["n"]), T), int(it["ld"])
        if rows == 0 or n == 0:
            continue
        dt = NP_DT[int(it["dtype"])]
        off = int(it["src"]) - base
        src = np.frombuffer(staging, dtype=dt, count=(rows - 1) * ld + n, offset=off)
        view = np.lib.stride_tricks.as_strided(src, (rows, n), (ld * src.itemsize, src.itemsize))
        out[b, :rows, :n] = view.astype(np.float32)
    return out


@pytest.mark.parametrize("modal_ch", [208, 273])
def test_staged_bytes_and_item_table_reproduce_the_collated_batch(tmp_path, modal_ch):
    proc = SyntheticProcessor(WHISPER_BASE)
    jl = write_cases(str(tmp_path), modal_ch)
    ds, ds_raw = datasets(jl, proc, modal_ch)
    coll = DataCollatorSpeechSeq2SeqWithPadding(processor=proc)
    ref = coll([ds[i] for i in range(len(ds))])
    raw = coll([ds_raw[i] for i in range(len(ds_raw))])
    assert np.array_equal(ref["labels"].numpy(), raw["labels"].numpy())
    raws = raw["input_features"]
    assert isinstance(raws, list) and all(isinstance(r, RawSignal) for r in raws)
    T = 6000
    plans = [plan_read(r, T) for r in raws]
    kinds = {p.kind for p in plans}
    assert kinds == {"span", "rows", "array"}
    offs, used = layout_batch(plans)
    assert all(o % 256 == 0 for o in offs) and used >= sum(p.nbytes for p in plans)
    staging = bytearray(used)
    mv = memoryview(staging)
    for p, o in zip(plans, offs):
        _fill(p, mv[o:o + p.nbytes])
    base = 0x7F0000000000
    table = item_table(plans, offs, base)
    assert table.dtype.itemsize == ITEM_BYTES and table.nbytes == ITEM_BYTES * len(plans)
    got = replay(table, staging, base, modal_ch, T)
    assert np.array_equal(got, ref["input_features"].numpy())
    # recordings longer than 2T never stage more than T samples per row
    long = [p for p in plans if p.kind == "rows"][0]
    assert long.n == T and long.nbytes == long.rows * T * long.itemsize


def test_header_parsing_and_shape_assertions(tmp_path):
    p = str(tmp_path / "a.npy")
    x = np.arange(12, dtype=np.float64).reshape(3, 4)
    np.save(p, x)
    off, shape, descr, fortran = npy_layout(p)
    assert shape == (3, 4) and descr == "<f8" and not fortran
    assert np.array_equal(np.fromfile(p, dtype=np.float64, offset=off), x.reshape(-1))
    # a slice with more rows than the model has channels is the reader's own shape assertion (reader.py:503-505)
    q = str(tmp_path / "schoffelen_b.npy")
    np.save(q, np.zeros((301, 10)))
    with pytest.raises(AssertionError):
        plan_read(RawSignal(q, 28, 301, 208), 6000)
    # 1-D file: pad_sample_ch's ndim assertion (reader.py:510)
    r = str(tmp_path / "c.npy")
    np.save(r, np.zeros(10))
    with pytest.raises(AssertionError):
        plan_read(RawSignal(r, 0, 208, 208), 6000)
    with pytest.raises(ValueError):
        open(str(tmp_path / "d.npy"), "wb").write(b"not numpy at all")
        npy_layout(str(tmp_path / "d.npy"))
    # slice starting beyond the file: zero rows, nothing staged
    pl = plan_read(RawSignal(p, 28, 301, 273), 6000)
    assert pl.rows == 0 and pl.nbytes == 0


def test_loader_workers_come_from_a_forkserver_and_do_not_reimport_the_callers_script(tmp_path):
    """DataLoader workers (reference: finetune.py:249, evaluation.py:126-127) are never forked from the process that drives the GPU
    (utils.data_utils.worker_context: forkserver; a default-context loader is refused), and starting them does not re-import the
    caller's main script: an UNGUARDED script that builds a loader -- what tools/run_recipe.py or a notebook cell calling
    finetune.main() is -- must run its body once, with persistent workers serving two epochs."""
    import subprocess
    import sys
    import textwrap
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    script = tmp_path / "unguarded_main.py"
    script.write_text(textwrap.dedent(f"""
        import sys, os
        sys.path.insert(0, {root!r})
        import torch
        from neuspeech1_amd.synthetic import SyntheticProcessor
        from neuspeech1_amd.weights import WHISPER_BASE
        from tests.feed_cases import datasets, write_cases
        from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding, worker_context, fork_safe_iter, start_worker_server
        print("MAIN BODY RUNS", flush=True)
        start_worker_server()
        proc = SyntheticProcessor(WHISPER_BASE)
        jl = write_cases({str(tmp_path)!r}, 208)
        ds, ds_raw = datasets(jl, proc, 208)
        coll = DataCollatorSpeechSeq2SeqWithPadding(processor=proc)
        loader = torch.utils.data.DataLoader(ds_raw, batch_size=4, num_workers=2, collate_fn=coll, persistent_workers=True,
                                             multiprocessing_context=worker_context(2))
        for ep in range(2):
            print("EPOCH", ep, sum(len(b["labels"]) for b in fork_safe_iter(loader)), flush=True)
        try:
            fork_safe_iter(torch.utils.data.DataLoader(ds, batch_size=4, num_workers=2, collate_fn=coll))
        except RuntimeError as e:
            print("REFUSED", str(e)[:40], flush=True)
        def my_collate(batch):       # defined in the main script: a worker that does not re-import it cannot unpickle this
            return coll(batch)
        try:
            fork_safe_iter(torch.utils.data.DataLoader(ds_raw, batch_size=4, num_workers=2, collate_fn=my_collate,
                                                       multiprocessing_context=worker_context(2)))
        except RuntimeError as e:
            print("LOCAL", "collate_fn (my_collate)" in str(e) and "importable module" in str(e), flush=True)
        print("FILE", __file__ is not None, flush=True)
    """))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = r.stdout
    assert out.count("MAIN BODY RUNS") == 1, out
    assert "EPOCH 0 10" in out and "EPOCH 1 10" in out and "REFUSED DataLoader workers must come from" in out and "FILE True" in out, out
    assert "LOCAL True" in out, out      # objects defined in the main script are named up front, not by a pickling error inside a worker
