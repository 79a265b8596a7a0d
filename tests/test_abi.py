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
    assert so.ns_version() == 2 and so.ns_last_error() is not None


def test_bad_arguments_fail_loudly_without_gpu():
    import ctypes as C
    from neuspeech1_amd import lib
    so = lib.load()
    d = lib.GemmDesc()
    assert so.ns_gemm(C.byref(d), None) == -1
    assert b"ns_gemm" in so.ns_last_error()
