"""CPU: the C-ABI library loads and exports every symbol include/neuspeech_hip.h declares (no compute calls)."""
import os
import re

def test_header_symbols_are_exported_and_bound():
    from neuspeech1_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "neuspeech_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ns_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    so = lib.load()
    for name in declared:
        assert hasattr(so, name), f"{name} declared in the header but not exported"
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    assert so.ns_version() == 3 and so.ns_last_error() is not None


def test_bad_arguments_fail_loudly_without_gpu():
    import ctypes as C
    from neuspeech1_amd import lib
    so = lib.load()
    d = lib.GemmDesc()
    assert so.ns_gemm(C.byref(d), None) == -1
    assert b"ns_gemm" in so.ns_last_error()


def test_attention_backward_workspace_rule():
    """ns_attn_bwd_workspace_bytes (host-only): > 0 exactly where a one-pass backward exists -- unmasked attention over >= 256 keys with
    >= 256 queries (fp32 dQ scratch, one 64 x 64 tile per step) or <= 64 queries (fp32 dQ slabs, one per group of key blocks)."""
    from neuspeech1_amd import lib
    so = lib.load()
    f = so.ns_attn_bwd_workspace_bytes
    assert f(64, 8, 1500, 1500, 0) == 64 * 8 * 24 * 64 * 64 * 4
    few = f(64, 8, 44, 1500, 0)
    assert few > 0 and few % (64 * 8 * 44 * 64 * 4) == 0 and few // (64 * 8 * 44 * 64 * 4) <= 6
    for args in ((64, 8, 1500, 1500, 1), (64, 8, 44, 44, 1), (64, 8, 44, 200, 0), (64, 8, 100, 1500, 0), (64, 8, 200, 1500, 0)):
        assert f(*args) == 0, args


def test_launch_list_recording_is_thread_local_and_launches_nothing():
    """ops.recording / ops.LaunchList (generate.py replays them for generations too short for hipGraphs): while a list is being recorded on
    a thread, entry points are appended with their marshalled arguments instead of being launched -- as a stream capture records instead of
    executing -- and ONLY on that thread (the data feed's loader thread launches ns_feed_pack while the main thread may be recording)."""
    import threading
    import torch
    from neuspeech1_amd import ops
    t = torch.zeros(4, dtype=torch.int32)
    lst = ops.LaunchList()
    seen = {}
    with ops.recording(lst):
        ops.add_i32(t, 3)                      # on a CPU tensor: launching this would fail (no GPU here); recording must not call it
        ops.zero_(torch.zeros(8))
        th = threading.Thread(target=lambda: seen.setdefault("rec", getattr(ops._tls, "rec", None)))
        th.start()
        th.join()
    assert seen["rec"] is None                # another thread does not see this thread's recording
    assert [c[1] for c in lst.calls] == ["ns_add_i32", "ns_zero_spans"]
    fn, name, args = lst.calls[0]
    assert args[0] == t.data_ptr() and args[1] == 1 and args[2] == 3     # (pointer, n, v)
    assert getattr(ops._tls, "rec", None) is None
    assert int(t[0]) == 0                     # nothing ran
