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
