"""Identity of a kernel's SOURCE for the PMC evidence files under profiles/: the sha256 of the named csrc files with
comments and whitespace removed, so that a comment edit does not orphan a rocprofv3 pass (round 3: a two-line comment
made bench.py refuse profiles/r3_pmc_traffic.json and the driver's line carried `traffic: null`).  bench.py reports a
counter file only while this hash equals the one recorded in it; tests/test_profiles_cpu.py fails when they differ."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neuspeech1_amd", "csrc")
# the dominant training kernel (bench.py: DOMINANT) and what its tile arithmetic / epilogues are written in
DOMINANT_SOURCES = ("ns_gemm_p8s.hip", "ns_gemm_p8.hip", "ns_gemm_epi.h")
# ... plus the dispatch rule that decides WHICH launches make up the class (moving NS_P8S_MIN_TILES from 1024 to 700 changed
# the class mix -- and with it the per-launch average -- without touching a hashed file): lines of ns_gemm.hip matching these patterns
DOMINANT_DISPATCH = ("ns_gemm.hip", (r"#define\s+NS_P8S_MIN_TILES\s+\d+", r"const bool big = [^;]*;", r"const bool pers = [^;]*;"))
# one decode step: what moves its bytes -- attention over the cross / self caches (ns_attn_fewq, ns_attn_decode: 79 % / 92 % of a
# beam-5 / greedy step's HBM traffic) and the small-M projections
DECODE_SOURCES = ("ns_decode.hip", "ns_gemm_smallm.hip")

_COMMENT = re.compile(r"//[^\n]*|/\*.*?\*/", re.S)


def stripped(text: str) -> str:
    """C / C++ comments and all whitespace removed (string literals in these files hold no comment markers)"""
    return re.sub(r"\s+", "", _COMMENT.sub(" ", text))


def source_hash(files=DOMINANT_SOURCES) -> str:
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(CSRC, f), encoding="utf-8") as fh:
            h.update(stripped(fh.read()).encode())
    if files is DOMINANT_SOURCES:
        name, pats = DOMINANT_DISPATCH
        with open(os.path.join(CSRC, name), encoding="utf-8") as fh:
            text = _COMMENT.sub(" ", fh.read())
        for p in pats:
            found = re.findall(p, text, re.S)
            assert found, f"{name}: dispatch pattern {p!r} not found (tools/kernel_hash.py is out of date)"
            h.update(stripped("".join(found)).encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print("dominant", source_hash(DOMINANT_SOURCES))
    print("decode", source_hash(DECODE_SOURCES))
