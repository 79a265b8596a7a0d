"""Build libneuspeech_hip.so (gfx950 only) in-tree with hipcc.

No cmake/ninja, no torch extension machinery: the library is a plain C-ABI
shared object (see include/neuspeech_hip.h) linked against libamdhip64 only.
Objects are cached by source mtime so rebuilds take seconds.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libneuspeech_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value",
         "-fno-gpu-rdc", "-ffp-contract=fast"]
FLAGS += os.environ.get("NS_EXTRA_HIPCC_FLAGS", "").split()   # diagnostic builds only (e.g. -DNS_P8_STAMPS)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "neuspeech_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    spath = os.path.join(CSRC, src)
    newest = max(os.path.getmtime(spath), _deps_mtime())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    cmd = [HIPCC, *FLAGS, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def stale_sources():
    """sources / headers newer than the built library (empty list: the .so is current)"""
    if not os.path.exists(LIB):
        return sources()
    t = os.path.getmtime(LIB)
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(HERE, "..", "include", "neuspeech_hip.h"))
    return [os.path.basename(f) for f in files if os.path.getmtime(f) > t]


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(BUILD, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    newest_obj = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest_obj:
        tmp = f"{LIB}.tmp.{os.getpid()}"    # link beside the target, then rename: no reader ever sees a partial ELF
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, LIB)
    if verbose:
        print(f"[neuspeech1_amd] built {LIB} from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
