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


def test_feed_cache_files_hold_the_collator_and_autocast_roundings(tmp_path):
    """neuspeech1_amd.feed.plan_cached (round 6, host logic only): the cache file of a recording holds round16(round32(x)) (f16) or round32(x)
    (f32) of exactly the rows / samples the plain plan would stage -- channel slice, crop of recordings longer than 2 T --, is reused by a
    second plan, is rebuilt when the source file changes, is skipped where nothing narrower exists (float16 sources, f32 sources with an f32
    cache, exotic dtypes that numpy loads), and two builders racing for one file both leave a complete file."""
    import os
    import threading
    import numpy as np
    from neuspeech1_amd.feed import NS_FEED_F16, NS_FEED_F32, RawSignal, _fill, cache_path, plan_cached, plan_read
    rng = np.random.default_rng(3)
    T, cdir = 600, str(tmp_path / "cache")
    a = rng.standard_normal((30, 500)) * 1e3
    a[0, :4] = [1e-8, 70000.0, -1e-30, 65504.0]          # fp16 subnormal / overflow / underflow / largest finite
    p = str(tmp_path / "a.npy")
    np.save(p, a)
    raw = RawSignal(p, 4, 24, 20)
    for dt, code, np_t in (("f16", NS_FEED_F16, np.float16), ("f32", NS_FEED_F32, np.float32)):
        pl = plan_cached(raw, T, cdir, dt)
        want = a[4:24].astype(np.float32)
        with np.errstate(over="ignore"):
            want = want.astype(np_t)
        got = np.load(pl.path)
        assert pl.kind == "span" and pl.dtype == code and (pl.rows, pl.n) == (20, 500) and got.dtype == np_t
        assert np.array_equal(got.view(np.uint16 if dt == "f16" else np.uint32), want.view(np.uint16 if dt == "f16" else np.uint32))
        buf = bytearray(pl.nbytes)
        _fill(pl, memoryview(buf))
        assert bytes(buf) == want.tobytes()
        m0 = os.stat(pl.path).st_mtime_ns
        assert plan_cached(raw, T, cdir, dt).path == pl.path and os.stat(pl.path).st_mtime_ns == m0      # reused, not rebuilt
    # a recording longer than 2 T: only the first T samples of each row are cached
    b = rng.standard_normal((20, 3 * T + 7))
    pb = str(tmp_path / "b.npy")
    np.save(pb, b)
    pl = plan_cached(RawSignal(pb, 0, 20, 20), T, cdir, "f16")
    assert np.array_equal(np.load(pl.path), b[:, :T].astype(np.float32).astype(np.float16))
    # the source changes: a new cache file (size / mtime are part of its name)
    old = cache_path(cdir, raw, T, "f16")
    np.save(p, a * 2.0)
    os.utime(p, ns=(os.stat(p).st_atime_ns, os.stat(p).st_mtime_ns + 5_000_000))
    pl2 = plan_cached(raw, T, cdir, "f16")
    assert pl2.path != old and np.array_equal(np.load(pl2.path), (a[4:24] * 2.0).astype(np.float32).astype(np.float16))
    # nothing narrower to store: the plain plan comes back
    h = str(tmp_path / "h.npy")
    np.save(h, rng.standard_normal((20, 100)).astype(np.float16))
    f = str(tmp_path / "f.npy")
    np.save(f, rng.standard_normal((20, 100)).astype(np.float32))
    i16 = str(tmp_path / "i.npy")
    np.save(i16, rng.integers(-5, 5, (20, 100)).astype(np.int16))
    assert plan_cached(RawSignal(h, 0, 20, 20), T, cdir, "f16").path == h
    assert plan_cached(RawSignal(f, 0, 20, 20), T, cdir, "f32").path == f
    assert plan_cached(RawSignal(f, 0, 20, 20), T, cdir, "f16").path != f
    assert plan_cached(RawSignal(i16, 0, 20, 20), T, cdir, "f16").kind == "array" == plan_read(RawSignal(i16, 0, 20, 20), T).kind
    # two ranks touch a new recording at once
    c = str(tmp_path / "c.npy")
    np.save(c, rng.standard_normal((20, 400)))
    res = []
    ths = [threading.Thread(target=lambda: res.append(plan_cached(RawSignal(c, 0, 20, 20), T, cdir, "f16"))) for _ in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert len({r.path for r in res}) == 1 and np.load(res[0].path).shape == (20, 400)
    assert not [f_ for f_ in os.listdir(cdir) if ".tmp." in f_]
